// Resize op for Scanner on MI355X.
//
// Drop-in for the reference's kernel
//   ResizeKernel  /root/reference/scannertools/scannertools_cpp/imgproc/resize_kernel.cpp:22-110
// Same op declaration (frame_input("frame") -> frame_output("frame"),
// stream_protobuf_name("ResizeArgs")), same per-stream arguments (ResizeArgs{width = 1, height = 2,
// min = 3, preserve_aspect = 4, interpolation = 5}, scannertools_imgproc.proto:33-39) and the same
// target-size rules (resize_kernel.cpp:44-62).  The per-frame cv::resize / cvc::resize calls are
// replaced by ONE st_resize_u8_batch() call per execute().  Implemented interpolations:
// INTER_LINEAR (the reference's default, :31), INTER_NEAREST, INTER_CUBIC, INTER_AREA and
// INTER_LANCZOS4; the other names of the reference's table (:10-19: INTER_MAX and the warp flags, which
// are not interpolation modes) are rejected when the
// stream starts instead of being run on the CPU.
#include <map>

#include "scanner/api/kernel.h"
#include "scanner/api/op.h"
#include "scanner/util/hip.h"
#include "scanner/util/memory.h"
#include "proto_lite.h"
#include "scannertools_hip.h"
#include "stage.h"

namespace scanner {
namespace {
// cv::InterpolationFlags values of the names the reference accepts (resize_kernel.cpp:10-19)
const std::map<std::string, int> INTERP_TYPES = {
    {u8"INTER_NEAREST", 0}, {u8"INTER_LINEAR", 1}, {u8"INTER_CUBIC", 2}, {u8"INTER_AREA", 3},
    {u8"INTER_LANCZOS4", 4}, {u8"INTER_MAX", 7}, {u8"WARP_FILL_OUTLIERS", 8}, {u8"WARP_INVERSE_MAP", 16},
};

struct ResizeArgsLite {
  i32 width = 0, height = 0;
  bool min = false, preserve_aspect = false;
  std::string interpolation;
};

bool parse_resize_args(const std::vector<u8>& args, ResizeArgsLite* out) {
  std::vector<proto_lite::Field> fields;
  if (!proto_lite::parse(args.data(), args.size(), &fields)) return false;
  *out = ResizeArgsLite();
  for (auto& f : fields) {
    if (f.number == 1 && f.wire == 0) out->width = (i32)f.value;
    if (f.number == 2 && f.wire == 0) out->height = (i32)f.value;
    if (f.number == 3 && f.wire == 0) out->min = f.value != 0;
    if (f.number == 4 && f.wire == 0) out->preserve_aspect = f.value != 0;
    if (f.number == 5 && f.wire == 2) out->interpolation = f.bytes;
  }
  return true;
}
}  // namespace

template <bool STAGED>
class ResizeKernelHIPImpl : public BatchedKernel {
 public:
  ResizeKernelHIPImpl(const KernelConfig& config)
    : BatchedKernel(config), device_(config.devices[0]), gpu_(STAGED ? staging_device_id() : config.devices[0].id),
      stage_(gpu_), pipe_(gpu_) {
    if (!STAGED && device_.type != DeviceType::GPU) {
      RESULT_ERROR(&valid_, "ResizeKernelHIP runs on DeviceType::GPU only");
      return;
    }
    int st = st_ctx_create(gpu_, &ctx_);
    if (st != ST_OK) {
      RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s (no CPU fallback exists)", gpu_, st_status_string(st));
    } else if (STAGED && (!pipe_.init() || st_ctx_set_stream(ctx_, pipe_.compute_stream()) != ST_OK)) {
      RESULT_ERROR(&valid_, "cannot create the upload pipeline on device %d", gpu_);
    }
  }
  ~ResizeKernelHIPImpl() {
    if (ctx_) st_ctx_destroy(ctx_);
  }
  void validate(Result* result) override {
    result->set_msg(valid_.msg());
    result->set_success(valid_.success());
  }

  void new_stream(const std::vector<u8>& args) override {
    LOG_IF(FATAL, !parse_resize_args(args, &args_)) << "Resize: could not parse ResizeArgs";
    interp_type_ = 1;  // cv::INTER_LINEAR (resize_kernel.cpp:31)
    if (INTERP_TYPES.count(args_.interpolation) > 0) interp_type_ = INTERP_TYPES.at(args_.interpolation);
    LOG_IF(FATAL, interp_type_ < ST_INTER_NEAREST || interp_type_ > ST_INTER_LANCZOS4)
        << "Resize: interpolation " << args_.interpolation << " is not implemented on this device "
        << "(INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA and INTER_LANCZOS4 are)";
  }

  void execute(const BatchedElements& input_columns, BatchedElements& output_columns) override {
    auto& frame_col = input_columns[0];
    i32 input_count = (i32)num_rows(frame_col);
    if (input_count == 0) return;
    const Frame* frame = frame_col[0].as_const_frame();
    LOG_IF(FATAL, frame->type != FrameType::U8) << "Resize expects U8 frames";

    // resize_kernel.cpp:44-62
    i32 target_width = args_.width;
    i32 target_height = args_.height;
    if (args_.preserve_aspect) {
      if (target_width == 0) {
        target_width = frame->width() * target_height / frame->height();
      } else {
        target_height = frame->height() * target_width / frame->width();
      }
    }
    if (args_.min) {
      if (frame->width() <= target_width && frame->height() <= target_height) {
        target_width = frame->width();
        target_height = frame->height();
      }
    }
    LOG_IF(FATAL, target_width <= 0 || target_height <= 0) << "Resize: empty target size";

    FrameInfo info(target_height, target_width, frame->channels(), frame->type);
    std::vector<Frame*> output_frames = new_frames(device_, info, input_count);
    src_.resize(input_count);
    dst_.resize(input_count);
    const size_t in_bytes = frame->size(), out_bytes = info.size();
    if (STAGED) {
      // Host frames: the uploads are the work (a 1080p frame is 110 us of PCIe for a few us of kernel), so they run back to
      // back on a copy stream in sub-batches that alternate between two device slots while the compute stream resizes the
      // sub-batch that has just arrived (stage.h: UploadPipeline, as the staged Histogram kernel); the resized frames are
      // packed on the device and come back in one copy.  (Was: one synchronous copy per frame each way, 0.73 of the H2D rate.)
      const size_t in_stride = DeviceStage::align(in_bytes);
      u8* dev_out = stage_.reserve(out_bytes * (size_t)input_count + 256);
      const i32 fh = frame->height(), fw = frame->width(), fc = frame->channels();
      for (i32 i = 0; i < input_count; ++i)
        LOG_IF(FATAL, frame_col[i].as_const_frame()->size() != in_bytes) << "Resize: frame " << i << " changes shape inside a batch";
      pipe_.run(input_count, 8, in_bytes, in_stride,
                [&](i32 i) { return (const u8*)frame_col[i].as_const_frame()->data; },
                [&](u8* dev, i32 first, i32 nb) {
                  for (i32 i = 0; i < nb; ++i) {
                    src_[first + i] = dev + in_stride * i;
                    dst_[first + i] = dev_out + out_bytes * (size_t)(first + i);
                  }
                  int st2 = st_resize_u8_batch(ctx_, src_.data() + first, nb, fh, fw, fc, target_height, target_width, interp_type_,
                                               dst_.data() + first);
                  LOG_IF(FATAL, st2 != ST_OK) << "st_resize_u8_batch: " << st_ctx_last_error(ctx_);
                });
      // one copy when the output frames are one host block (new_frames), else frame by frame
      bool packed = true;
      for (i32 i = 1; i < input_count; ++i) packed = packed && output_frames[i]->data == output_frames[i - 1]->data + out_bytes;
      if (packed) {
        HIP_CHECK(hipMemcpyAsync(output_frames[0]->data, dev_out, out_bytes * (size_t)input_count, hipMemcpyDeviceToHost, pipe_.compute_stream()));
      } else {
        for (i32 i = 0; i < input_count; ++i)
          HIP_CHECK(hipMemcpyAsync(output_frames[i]->data, dst_[i], out_bytes, hipMemcpyDeviceToHost, pipe_.compute_stream()));
      }
      pipe_.drain();
      for (i32 i = 0; i < input_count; ++i) insert_frame(output_columns[0], output_frames[i]);
      return;
    } else {
      for (i32 i = 0; i < input_count; ++i) {
        src_[i] = frame_col[i].as_const_frame()->data;
        dst_[i] = output_frames[i]->data;
      }
    }
    int st = st_resize_u8_batch(ctx_, src_.data(), input_count, frame->height(), frame->width(), frame->channels(),
                                target_height, target_width, interp_type_, dst_.data());
    LOG_IF(FATAL, st != ST_OK) << "st_resize_u8_batch: " << st_ctx_last_error(ctx_);
    st = st_ctx_sync(ctx_);
    LOG_IF(FATAL, st != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(ctx_);
    for (i32 i = 0; i < input_count; ++i) insert_frame(output_columns[0], output_frames[i]);
  }

 private:
  DeviceHandle device_;
  int gpu_;
  DeviceStage stage_;
  UploadPipeline pipe_;
  ResizeArgsLite args_;
  int interp_type_ = 1;
  Result valid_;
  st_ctx* ctx_ = nullptr;
  std::vector<const uint8_t*> src_;
  std::vector<uint8_t*> dst_;
};

using ResizeKernelHIP = ResizeKernelHIPImpl<false>;
using ResizeKernelHIPStaged = ResizeKernelHIPImpl<true>;

REGISTER_OP(Resize).frame_input("frame").frame_output("frame").stream_protobuf_name("ResizeArgs");

REGISTER_KERNEL(Resize, ResizeKernelHIPStaged).device(DeviceType::CPU).batch().num_devices(1);

REGISTER_KERNEL(Resize, ResizeKernelHIP).device(DeviceType::GPU).batch().num_devices(1);
}
