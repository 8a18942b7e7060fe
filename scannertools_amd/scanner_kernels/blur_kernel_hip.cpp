// Blur op for Scanner on MI355X.
//
// Drop-in for the reference's kernel
//   BlurKernel  /root/reference/scannertools/scannertools_cpp/imgproc/blur_kernel_cpu.cpp:25-99
// Same op declaration (frame_input("frame") -> frame_output("frame"), protobuf_name("BlurArgs")),
// same arguments (BlurArgs{kernel_size = 1, sigma = 2}; sigma is parsed and unused, as in the
// reference), same output frame shape and type.  The reference's per-pixel k x k loop is replaced
// by ONE st_box_blur_u8c3_batch() call per execute().  Differences a caller can observe: border
// pixels, which the reference leaves uninitialised, are 0; the kernels are registered batched
// (Scanner may hand them several rows at once).  The reference registers the op on
// DeviceType::CPU only; that registration stages host frames through the GPU.
#include "scanner/api/kernel.h"
#include "scanner/api/op.h"
#include "scanner/util/hip.h"
#include "scanner/util/memory.h"
#include "proto_lite.h"
#include "scannertools_hip.h"
#include "stage.h"

namespace scanner {
namespace {
// BlurArgs (scannertools_imgproc.proto:3-6): int32 kernel_size = 1; float sigma = 2;
bool parse_blur_args(const std::vector<u8>& args, i32* kernel_size, f32* sigma) {
  std::vector<proto_lite::Field> fields;
  if (args.empty() || !proto_lite::parse(args.data(), args.size(), &fields)) return false;
  *kernel_size = 0;
  *sigma = 0.f;
  for (auto& f : fields) {
    if (f.number == 1 && f.wire == 0) *kernel_size = (i32)f.value;
    if (f.number == 2 && f.wire == 5) *sigma = proto_lite::as_float(f);
  }
  return true;
}
}  // namespace

template <bool STAGED>
class BlurKernelHIPImpl : public BatchedKernel, public VideoKernel {
 public:
  BlurKernelHIPImpl(const KernelConfig& config)
    : BatchedKernel(config), device_(config.devices[0]), gpu_(STAGED ? staging_device_id() : config.devices[0].id),
      stage_(gpu_) {
    // blur_kernel_cpu.cpp:29-33: an empty or unparsable BlurArgs invalidates the kernel
    if (!parse_blur_args(config.args, &kernel_size_, &sigma_)) {
      RESULT_ERROR(&valid_, "Could not parse BlurArgs");
      return;
    }
    if (!STAGED && device_.type != DeviceType::GPU) {
      RESULT_ERROR(&valid_, "BlurKernelHIP runs on DeviceType::GPU only");
      return;
    }
    if (kernel_size_ < 1 || kernel_size_ > 31) {
      RESULT_ERROR(&valid_, "Blur kernel_size must be in [1, 31], got %d", kernel_size_);
      return;
    }
    int st = st_ctx_create(gpu_, &ctx_);
    if (st != ST_OK) RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s (no CPU fallback exists)", gpu_, st_status_string(st));
  }
  ~BlurKernelHIPImpl() {
    if (ctx_) st_ctx_destroy(ctx_);
  }
  void validate(Result* result) override {
    result->set_msg(valid_.msg());
    result->set_success(valid_.success());
  }

  void execute(const BatchedElements& input_columns, BatchedElements& output_columns) override {
    auto& frame_col = input_columns[0];
    i32 input_count = (i32)num_rows(frame_col);
    if (input_count == 0) return;
    check_frame(device_, frame_col[0]);
    LOG_IF(FATAL, frame_info_.channels() != 3 || frame_info_.type != FrameType::U8)
        << "Blur expects U8 frames with 3 channels";
    const i32 h = frame_info_.height(), w = frame_info_.width();
    FrameInfo info = frame_col[0].as_const_frame()->as_frame_info();  // blur_kernel_cpu.cpp:58
    std::vector<Frame*> output_frames = new_frames(device_, info, input_count);
    src_.resize(input_count);
    dst_.resize(input_count);
    if (STAGED) {
      const size_t frame_bytes = frame_info_.size(), stride = DeviceStage::align(frame_bytes);
      u8* dev = stage_.reserve(2 * stride * input_count);
      for (i32 i = 0; i < input_count; ++i) {
        stage_.upload(dev + stride * i, frame_col[i].as_const_frame()->data, frame_bytes);
        src_[i] = dev + stride * i;
        dst_[i] = dev + stride * (input_count + i);
      }
      run(input_count, h, w);
      for (i32 i = 0; i < input_count; ++i) stage_.download(output_frames[i]->data, dst_[i], frame_bytes);
    } else {
      for (i32 i = 0; i < input_count; ++i) {
        src_[i] = frame_col[i].as_const_frame()->data;
        dst_[i] = output_frames[i]->data;
      }
      run(input_count, h, w);
    }
    for (i32 i = 0; i < input_count; ++i) insert_frame(output_columns[0], output_frames[i]);
  }

 private:
  void run(i32 n, i32 h, i32 w) {
    int st = st_box_blur_u8c3_batch(ctx_, src_.data(), n, h, w, kernel_size_, dst_.data());
    LOG_IF(FATAL, st != ST_OK) << "st_box_blur_u8c3_batch: " << st_ctx_last_error(ctx_);
    st = st_ctx_sync(ctx_);
    LOG_IF(FATAL, st != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(ctx_);
  }

  DeviceHandle device_;
  int gpu_;
  DeviceStage stage_;
  i32 kernel_size_ = 0;
  f32 sigma_ = 0.f;
  Result valid_;
  st_ctx* ctx_ = nullptr;
  std::vector<const uint8_t*> src_;
  std::vector<uint8_t*> dst_;
};

using BlurKernelHIP = BlurKernelHIPImpl<false>;
using BlurKernelHIPStaged = BlurKernelHIPImpl<true>;

REGISTER_OP(Blur).frame_input("frame").frame_output("frame").protobuf_name("BlurArgs");

REGISTER_KERNEL(Blur, BlurKernelHIPStaged).device(DeviceType::CPU).batch().num_devices(1);

REGISTER_KERNEL(Blur, BlurKernelHIP).device(DeviceType::GPU).batch().num_devices(1);
}
