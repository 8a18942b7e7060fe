// ConvertColor op for Scanner on MI355X.
//
// Drop-in for the reference's kernel
//   ConvertColorKernel  /root/reference/scannertools/scannertools_cpp/imgproc/convert_color_kernel.cpp:213-309
// Same op declaration (frame_input("frame") -> frame_output("frame"),
// stream_protobuf_name("ConvertColorArgs")), same per-stream argument (ConvertColorArgs{conversion = 1},
// scannertools_imgproc.proto:29-31: the name of a cv::ColorConversionCodes constant).  The per-frame
// cv::cvtColor / cvc::cvtColor calls are replaced by ONE st_cvt_color_u8_batch() call per execute().
// Implemented names: COLOR_BGR2RGB, COLOR_RGB2BGR, COLOR_BGR2GRAY, COLOR_RGB2GRAY, COLOR_GRAY2BGR,
// COLOR_GRAY2RGB, COLOR_BGR/RGB2YCrCb, COLOR_YCrCb2BGR/RGB, COLOR_BGR/RGB2YUV, COLOR_YUV2BGR/RGB,
// COLOR_BGR/RGB2HSV, COLOR_HSV2BGR/RGB and the four _FULL hue-range variants, the same eight names for HLS, XYZ, the YUV 4:2:0 /
// 4:2:2 sources, and the channel layout family (codes 0..3, 5,
// 9..31: BGRA / RGBA, BGR565, BGR555, gray from / to them); an unknown name invalidates the stream as in the reference
// (:231-236), a name of the reference's table that is not implemented here is reported the same way
// instead of being run on the CPU.  SCANNERTOOLS_GRAY_BITS (15 default, 14) selects the luma table
// width of the OpenCV build being replaced.
#include <cstdlib>
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
const std::map<std::string, int> COLOR_CONVERSION_TYPES = {
    {u8"COLOR_BGR2RGB", ST_COLOR_BGR2RGB},   {u8"COLOR_RGB2BGR", ST_COLOR_RGB2BGR},
    {u8"COLOR_BGR2GRAY", ST_COLOR_BGR2GRAY}, {u8"COLOR_RGB2GRAY", ST_COLOR_RGB2GRAY},
    {u8"COLOR_GRAY2BGR", ST_COLOR_GRAY2BGR}, {u8"COLOR_GRAY2RGB", ST_COLOR_GRAY2RGB},
    {u8"COLOR_BGR2YCrCb", ST_COLOR_BGR2YCrCb}, {u8"COLOR_RGB2YCrCb", ST_COLOR_RGB2YCrCb},
    {u8"COLOR_YCrCb2BGR", ST_COLOR_YCrCb2BGR}, {u8"COLOR_YCrCb2RGB", ST_COLOR_YCrCb2RGB},
    {u8"COLOR_BGR2HSV", ST_COLOR_BGR2HSV},   {u8"COLOR_RGB2HSV", ST_COLOR_RGB2HSV},
    {u8"COLOR_HSV2BGR", ST_COLOR_HSV2BGR},   {u8"COLOR_HSV2RGB", ST_COLOR_HSV2RGB},
    {u8"COLOR_BGR2HSV_FULL", ST_COLOR_BGR2HSV_FULL}, {u8"COLOR_RGB2HSV_FULL", ST_COLOR_RGB2HSV_FULL},
    {u8"COLOR_HSV2BGR_FULL", ST_COLOR_HSV2BGR_FULL}, {u8"COLOR_HSV2RGB_FULL", ST_COLOR_HSV2RGB_FULL},
    {u8"COLOR_BGR2HLS", ST_COLOR_BGR2HLS},   {u8"COLOR_RGB2HLS", ST_COLOR_RGB2HLS},
    {u8"COLOR_HLS2BGR", ST_COLOR_HLS2BGR},   {u8"COLOR_HLS2RGB", ST_COLOR_HLS2RGB},
    {u8"COLOR_BGR2HLS_FULL", ST_COLOR_BGR2HLS_FULL}, {u8"COLOR_RGB2HLS_FULL", ST_COLOR_RGB2HLS_FULL},
    {u8"COLOR_HLS2BGR_FULL", ST_COLOR_HLS2BGR_FULL}, {u8"COLOR_HLS2RGB_FULL", ST_COLOR_HLS2RGB_FULL},
    {u8"COLOR_BGR2YUV", ST_COLOR_BGR2YUV},   {u8"COLOR_RGB2YUV", ST_COLOR_RGB2YUV},
    {u8"COLOR_YUV2BGR", ST_COLOR_YUV2BGR},   {u8"COLOR_YUV2RGB", ST_COLOR_YUV2RGB},
    {u8"COLOR_BGR2XYZ", ST_COLOR_BGR2XYZ},   {u8"COLOR_RGB2XYZ", ST_COLOR_RGB2XYZ},
    {u8"COLOR_XYZ2BGR", ST_COLOR_XYZ2BGR},   {u8"COLOR_XYZ2RGB", ST_COLOR_XYZ2RGB},
    // YUV 4:2:0 (single-channel (3H/2, W) frames) and packed 4:2:2 ((H, W, 2) frames) sources; cv::ColorConversionCodes values
    {u8"COLOR_YUV2RGB_NV12", 90},
    {u8"COLOR_YUV2BGR_NV12", 91},
    {u8"COLOR_YUV2RGB_NV21", 92},
    {u8"COLOR_YUV2BGR_NV21", 93},
    {u8"COLOR_YUV420sp2RGB", 92},
    {u8"COLOR_YUV420sp2BGR", 93},
    {u8"COLOR_YUV2RGBA_NV12", 94},
    {u8"COLOR_YUV2BGRA_NV12", 95},
    {u8"COLOR_YUV2RGBA_NV21", 96},
    {u8"COLOR_YUV2BGRA_NV21", 97},
    {u8"COLOR_YUV420sp2RGBA", 96},
    {u8"COLOR_YUV420sp2BGRA", 97},
    {u8"COLOR_YUV2RGB_YV12", 98},
    {u8"COLOR_YUV2BGR_YV12", 99},
    {u8"COLOR_YUV2RGB_IYUV", 100},
    {u8"COLOR_YUV2BGR_IYUV", 101},
    {u8"COLOR_YUV2RGB_I420", 100},
    {u8"COLOR_YUV2BGR_I420", 101},
    {u8"COLOR_YUV420p2RGB", 98},
    {u8"COLOR_YUV420p2BGR", 99},
    {u8"COLOR_YUV2RGBA_YV12", 102},
    {u8"COLOR_YUV2BGRA_YV12", 103},
    {u8"COLOR_YUV2RGBA_IYUV", 104},
    {u8"COLOR_YUV2BGRA_IYUV", 105},
    {u8"COLOR_YUV2RGBA_I420", 104},
    {u8"COLOR_YUV2BGRA_I420", 105},
    {u8"COLOR_YUV420p2RGBA", 102},
    {u8"COLOR_YUV420p2BGRA", 103},
    {u8"COLOR_YUV2GRAY_420", 106},
    {u8"COLOR_YUV2GRAY_NV21", 106},
    {u8"COLOR_YUV2GRAY_NV12", 106},
    {u8"COLOR_YUV2GRAY_YV12", 106},
    {u8"COLOR_YUV2GRAY_IYUV", 106},
    {u8"COLOR_YUV2GRAY_I420", 106},
    {u8"COLOR_YUV420sp2GRAY", 106},
    {u8"COLOR_YUV420p2GRAY", 106},
    {u8"COLOR_YUV2RGB_UYVY", 107},
    {u8"COLOR_YUV2BGR_UYVY", 108},
    {u8"COLOR_YUV2RGB_Y422", 107},
    {u8"COLOR_YUV2BGR_Y422", 108},
    {u8"COLOR_YUV2RGB_UYNV", 107},
    {u8"COLOR_YUV2BGR_UYNV", 108},
    {u8"COLOR_YUV2RGBA_UYVY", 111},
    {u8"COLOR_YUV2BGRA_UYVY", 112},
    {u8"COLOR_YUV2RGBA_Y422", 111},
    {u8"COLOR_YUV2BGRA_Y422", 112},
    {u8"COLOR_YUV2RGBA_UYNV", 111},
    {u8"COLOR_YUV2BGRA_UYNV", 112},
    {u8"COLOR_YUV2RGB_YUY2", 115},
    {u8"COLOR_YUV2BGR_YUY2", 116},
    {u8"COLOR_YUV2RGB_YVYU", 117},
    {u8"COLOR_YUV2BGR_YVYU", 118},
    {u8"COLOR_YUV2RGB_YUYV", 115},
    {u8"COLOR_YUV2BGR_YUYV", 116},
    {u8"COLOR_YUV2RGB_YUNV", 115},
    {u8"COLOR_YUV2BGR_YUNV", 116},
    {u8"COLOR_YUV2RGBA_YUY2", 119},
    {u8"COLOR_YUV2BGRA_YUY2", 120},
    {u8"COLOR_YUV2RGBA_YVYU", 121},
    {u8"COLOR_YUV2BGRA_YVYU", 122},
    {u8"COLOR_YUV2RGBA_YUYV", 119},
    {u8"COLOR_YUV2BGRA_YUYV", 120},
    {u8"COLOR_YUV2RGBA_YUNV", 119},
    {u8"COLOR_YUV2BGRA_YUNV", 120},
    {u8"COLOR_YUV2GRAY_UYVY", 123},
    {u8"COLOR_YUV2GRAY_YUY2", 124},
    {u8"COLOR_YUV2GRAY_Y422", 123},
    {u8"COLOR_YUV2GRAY_UYNV", 123},
    {u8"COLOR_YUV2GRAY_YVYU", 124},
    {u8"COLOR_YUV2GRAY_YUYV", 124},
    {u8"COLOR_YUV2GRAY_YUNV", 124},
    // channel layout family: alpha channel added / dropped / swapped, 16-bit packed pixels (2-channel frames)
    {u8"COLOR_BGR2BGRA", ST_COLOR_BGR2BGRA}, {u8"COLOR_RGB2RGBA", ST_COLOR_BGR2BGRA}, {u8"COLOR_BGRA2BGR", ST_COLOR_BGRA2BGR},
    {u8"COLOR_RGBA2RGB", ST_COLOR_BGRA2BGR}, {u8"COLOR_BGR2RGBA", ST_COLOR_BGR2RGBA}, {u8"COLOR_RGB2BGRA", ST_COLOR_BGR2RGBA},
    {u8"COLOR_RGBA2BGR", ST_COLOR_RGBA2BGR}, {u8"COLOR_BGRA2RGB", ST_COLOR_RGBA2BGR}, {u8"COLOR_BGRA2RGBA", ST_COLOR_BGRA2RGBA},
    {u8"COLOR_RGBA2BGRA", ST_COLOR_BGRA2RGBA}, {u8"COLOR_GRAY2BGRA", ST_COLOR_GRAY2BGRA}, {u8"COLOR_GRAY2RGBA", ST_COLOR_GRAY2BGRA},
    {u8"COLOR_BGRA2GRAY", ST_COLOR_BGRA2GRAY}, {u8"COLOR_RGBA2GRAY", ST_COLOR_RGBA2GRAY},
    {u8"COLOR_BGR2BGR565", ST_COLOR_BGR2BGR565}, {u8"COLOR_RGB2BGR565", ST_COLOR_RGB2BGR565}, {u8"COLOR_BGR5652BGR", ST_COLOR_BGR5652BGR},
    {u8"COLOR_BGR5652RGB", ST_COLOR_BGR5652RGB}, {u8"COLOR_BGRA2BGR565", ST_COLOR_BGRA2BGR565}, {u8"COLOR_RGBA2BGR565", ST_COLOR_RGBA2BGR565},
    {u8"COLOR_BGR5652BGRA", ST_COLOR_BGR5652BGRA}, {u8"COLOR_BGR5652RGBA", ST_COLOR_BGR5652RGBA}, {u8"COLOR_GRAY2BGR565", ST_COLOR_GRAY2BGR565},
    {u8"COLOR_BGR5652GRAY", ST_COLOR_BGR5652GRAY}, {u8"COLOR_BGR2BGR555", ST_COLOR_BGR2BGR555}, {u8"COLOR_RGB2BGR555", ST_COLOR_RGB2BGR555},
    {u8"COLOR_BGR5552BGR", ST_COLOR_BGR5552BGR}, {u8"COLOR_BGR5552RGB", ST_COLOR_BGR5552RGB}, {u8"COLOR_BGRA2BGR555", ST_COLOR_BGRA2BGR555},
    {u8"COLOR_RGBA2BGR555", ST_COLOR_RGBA2BGR555}, {u8"COLOR_BGR5552BGRA", ST_COLOR_BGR5552BGRA}, {u8"COLOR_BGR5552RGBA", ST_COLOR_BGR5552RGBA},
    {u8"COLOR_GRAY2BGR555", ST_COLOR_GRAY2BGR555}, {u8"COLOR_BGR5552GRAY", ST_COLOR_BGR5552GRAY},
};
}

template <bool STAGED>
class ConvertColorKernelHIPImpl : public BatchedKernel {
 public:
  ConvertColorKernelHIPImpl(const KernelConfig& config)
    : BatchedKernel(config), device_(config.devices[0]), gpu_(STAGED ? staging_device_id() : config.devices[0].id),
      stage_(gpu_) {
    valid_.set_success(true);
    const char* gb = getenv("SCANNERTOOLS_GRAY_BITS");
    gray_bits_ = gb && atoi(gb) == 14 ? 14 : 15;
    if (!STAGED && device_.type != DeviceType::GPU) {
      RESULT_ERROR(&valid_, "ConvertColorKernelHIP runs on DeviceType::GPU only");
      return;
    }
    int st = st_ctx_create(gpu_, &ctx_);
    if (st != ST_OK) RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s (no CPU fallback exists)", gpu_, st_status_string(st));
  }
  ~ConvertColorKernelHIPImpl() {
    if (ctx_) st_ctx_destroy(ctx_);
  }
  void validate(Result* result) override {
    result->set_msg(valid_.msg());
    result->set_success(valid_.success());
  }

  void new_stream(const std::vector<u8>& args) override {
    std::vector<proto_lite::Field> fields;
    std::string conversion;
    if (proto_lite::parse(args.data(), args.size(), &fields))
      for (auto& f : fields)
        if (f.number == 1 && f.wire == 2) conversion = f.bytes;
    if (COLOR_CONVERSION_TYPES.count(conversion) > 0) {
      code_ = COLOR_CONVERSION_TYPES.at(conversion);
    } else {
      // convert_color_kernel.cpp:231-236
      std::string err = "ConvertColor: invalid color conversion argument provided: " + conversion;
      RESULT_ERROR(&valid_, "%s", err.c_str());
      code_ = -1;
    }
  }

  void execute(const BatchedElements& input_columns, BatchedElements& output_columns) override {
    auto& frame_col = input_columns[0];
    i32 input_count = (i32)num_rows(frame_col);
    if (input_count == 0) return;
    LOG_IF(FATAL, code_ < 0) << valid_.msg();
    const Frame* frame = frame_col[0].as_const_frame();
    LOG_IF(FATAL, frame->type != FrameType::U8) << "ConvertColor expects U8 frames";
    int out_h = 0, out_w = 0, out_channels = 0;
    LOG_IF(FATAL, st_cvt_color_out_shape(code_, frame->height(), frame->width(), frame->channels(), &out_h, &out_w, &out_channels) != 0)
        << "ConvertColor: conversion " << code_ << " does not apply to " << frame->width() << "x" << frame->height() << " frames of "
        << frame->channels() << " channel(s)";
    FrameInfo info(out_h, out_w, out_channels, FrameType::U8);
    std::vector<Frame*> output_frames = new_frames(device_, info, input_count);
    src_.resize(input_count);
    dst_.resize(input_count);
    const size_t in_bytes = frame->size(), out_bytes = info.size();
    if (STAGED) {
      const size_t in_stride = DeviceStage::align(in_bytes), out_stride = DeviceStage::align(out_bytes);
      u8* dev = stage_.reserve((in_stride + out_stride) * input_count);
      for (i32 i = 0; i < input_count; ++i) {
        stage_.upload(dev + in_stride * i, frame_col[i].as_const_frame()->data, in_bytes);
        src_[i] = dev + in_stride * i;
        dst_[i] = dev + in_stride * input_count + out_stride * i;
      }
    } else {
      for (i32 i = 0; i < input_count; ++i) {
        src_[i] = frame_col[i].as_const_frame()->data;
        dst_[i] = output_frames[i]->data;
      }
    }
    int st = st_cvt_color_u8_batch(ctx_, src_.data(), input_count, frame->height(), frame->width(), frame->channels(),
                                   code_, gray_bits_, dst_.data());
    LOG_IF(FATAL, st != ST_OK) << "st_cvt_color_u8_batch: " << st_ctx_last_error(ctx_);
    st = st_ctx_sync(ctx_);
    LOG_IF(FATAL, st != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(ctx_);
    if (STAGED)
      for (i32 i = 0; i < input_count; ++i) stage_.download(output_frames[i]->data, dst_[i], out_bytes);
    for (i32 i = 0; i < input_count; ++i) insert_frame(output_columns[0], output_frames[i]);
  }

 private:
  DeviceHandle device_;
  int gpu_;
  DeviceStage stage_;
  int code_ = -1;
  int gray_bits_ = 15;
  Result valid_;
  st_ctx* ctx_ = nullptr;
  std::vector<const uint8_t*> src_;
  std::vector<uint8_t*> dst_;
};

using ConvertColorKernelHIP = ConvertColorKernelHIPImpl<false>;
using ConvertColorKernelHIPStaged = ConvertColorKernelHIPImpl<true>;

REGISTER_OP(ConvertColor).frame_input("frame").frame_output("frame").stream_protobuf_name("ConvertColorArgs");

REGISTER_KERNEL(ConvertColor, ConvertColorKernelHIPStaged).device(DeviceType::CPU).batch().num_devices(1);

REGISTER_KERNEL(ConvertColor, ConvertColorKernelHIP).device(DeviceType::GPU).batch().num_devices(1);
}
