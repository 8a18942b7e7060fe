// CPM2Input op for Scanner on MI355X (pose path, BASELINE config 5).
//
// Drop-in for the reference's kernel
//   CPM2InputKernel  /root/reference/scannertools_caffe/scannertools_caffe_cpp/cpm2_input_kernel_gpu.cpp:26-187
// Same op declaration (frame_input("frame") -> frame_output("cpm2_input")), same arguments (CPM2Args{caffe_args
// = 1, scale = 2}, scannertools_caffe.proto:45-48; only `scale` is used, as in the reference), same output
// frame: FrameInfo(3, net_h, net_w, F32), planes B, G, R of the resized, padded, (x/256 - 0.5)-scaled frame.
// The reference issues six OpenCV-CUDA calls and a 2-D copy per frame on one of 32 streams; here ONE
// st_cpm2_input_batch() call covers the whole batch.  The reference registers the op on DeviceType::GPU
// only; a DeviceType::CPU registration (host frames staged through the GPU) is added, as for the imgproc ops.
#include "scanner/api/kernel.h"
#include "scanner/api/op.h"
#include "scanner/util/hip.h"
#include "scanner/util/memory.h"
#include "proto_lite.h"
#include "scannertools_hip.h"
#include "stage.h"

namespace scanner {
namespace {
// CPM2Args (scannertools_caffe.proto:45-48): CaffeArgs caffe_args = 1; float scale = 2;
bool parse_cpm2_scale(const std::vector<u8>& args, f32* scale) {
  std::vector<proto_lite::Field> fields;
  *scale = 0.f;
  if (!proto_lite::parse(args.data(), args.size(), &fields)) return false;
  for (auto& f : fields)
    if (f.number == 2 && f.wire == 5) *scale = proto_lite::as_float(f);
  return true;
}
}  // namespace

template <bool STAGED>
class CPM2InputKernelHIPImpl : public BatchedKernel, public VideoKernel {
 public:
  CPM2InputKernelHIPImpl(const KernelConfig& config)
    : BatchedKernel(config), device_(config.devices[0]), gpu_(STAGED ? staging_device_id() : config.devices[0].id),
      stage_(gpu_) {
    if (!parse_cpm2_scale(config.args, &scale_)) {
      RESULT_ERROR(&valid_, "Could not parse CPM2Args");
      return;
    }
    if (!(scale_ > 0.f)) {
      RESULT_ERROR(&valid_, "CPM2Input: scale must be positive, got %f", scale_);
      return;
    }
    if (!STAGED && device_.type != DeviceType::GPU) {
      RESULT_ERROR(&valid_, "CPM2InputKernelHIP runs on DeviceType::GPU only");
      return;
    }
    int st = st_ctx_create(gpu_, &ctx_);
    if (st != ST_OK) RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s (no CPU fallback exists)", gpu_, st_status_string(st));
  }
  ~CPM2InputKernelHIPImpl() {
    if (ctx_) st_ctx_destroy(ctx_);
  }
  void validate(Result* result) override {
    result->set_msg(valid_.msg());
    result->set_success(valid_.success());
  }

  void new_frame_info() override {
    // cpm2_input_kernel_gpu.cpp:44-55
    int st = st_cpm2_geometry(frame_info_.height(), frame_info_.width(), scale_, &resize_height_, &resize_width_,
                              &net_input_height_, &net_input_width_);
    LOG_IF(FATAL, st != ST_OK) << "CPM2Input: frame " << frame_info_.width() << "x" << frame_info_.height()
                               << " at scale " << scale_ << " gives an empty network input";
  }

  void execute(const BatchedElements& input_columns, BatchedElements& output_columns) override {
    auto& frame_col = input_columns[0];
    i32 input_count = (i32)num_rows(frame_col);
    if (input_count == 0) return;
    const auto eval_start = now();  // cpm2_input_kernel_gpu.cpp:92
    check_frame(device_, frame_col[0]);
    LOG_IF(FATAL, frame_info_.channels() != 3 || frame_info_.type != FrameType::U8)
        << "CPM2Input expects U8 frames with 3 channels";
    FrameInfo net_input_info(3, net_input_height_, net_input_width_, FrameType::F32);
    std::vector<Frame*> output_frames = new_frames(device_, net_input_info, input_count);
    src_.resize(input_count);
    dst_.resize(input_count);
    const size_t in_bytes = frame_info_.size(), out_bytes = net_input_info.size();
    if (STAGED) {
      const size_t in_stride = DeviceStage::align(in_bytes), out_stride = DeviceStage::align(out_bytes);
      u8* dev = stage_.reserve((in_stride + out_stride) * input_count);
      for (i32 i = 0; i < input_count; ++i) {
        LOG_IF(FATAL, frame_col[i].as_const_frame()->as_frame_info() != frame_info_) << "CPM2Input: frame shape changes inside a batch";
        stage_.upload(dev + in_stride * i, frame_col[i].as_const_frame()->data, in_bytes);
        src_[i] = dev + in_stride * i;
        dst_[i] = (float*)(dev + in_stride * input_count + out_stride * i);
      }
    } else {
      for (i32 i = 0; i < input_count; ++i) {
        LOG_IF(FATAL, frame_col[i].as_const_frame()->as_frame_info() != frame_info_) << "CPM2Input: frame shape changes inside a batch";
        src_[i] = frame_col[i].as_const_frame()->data;
        dst_[i] = (float*)output_frames[i]->data;
      }
    }
    int st = st_cpm2_input_batch(ctx_, src_.data(), input_count, frame_info_.height(), frame_info_.width(), scale_, dst_.data());
    LOG_IF(FATAL, st != ST_OK) << "st_cpm2_input_batch: " << st_ctx_last_error(ctx_);
    st = st_ctx_sync(ctx_);
    LOG_IF(FATAL, st != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(ctx_);
    if (STAGED)
      for (i32 i = 0; i < input_count; ++i) stage_.download(output_frames[i]->data, (const u8*)dst_[i], out_bytes);
    for (i32 i = 0; i < input_count; ++i) insert_frame(output_columns[0], output_frames[i]);
    if (profiler_) profiler_->add_interval("cpm2_input", eval_start, now());  // cpm2_input_kernel_gpu.cpp:153-155
  }

 private:
  DeviceHandle device_;
  int gpu_;
  DeviceStage stage_;
  f32 scale_ = 0.f;
  int resize_width_ = 0, resize_height_ = 0, net_input_width_ = 0, net_input_height_ = 0;
  Result valid_;
  st_ctx* ctx_ = nullptr;
  std::vector<const uint8_t*> src_;
  std::vector<float*> dst_;
};

using CPM2InputKernelHIP = CPM2InputKernelHIPImpl<false>;
using CPM2InputKernelHIPStaged = CPM2InputKernelHIPImpl<true>;

REGISTER_OP(CPM2Input).frame_input("frame").frame_output("cpm2_input").protobuf_name("CPM2Args");

REGISTER_KERNEL(CPM2Input, CPM2InputKernelHIP).device(DeviceType::GPU).batch().num_devices(1);

REGISTER_KERNEL(CPM2Input, CPM2InputKernelHIPStaged).device(DeviceType::CPU).batch().num_devices(1);
}
