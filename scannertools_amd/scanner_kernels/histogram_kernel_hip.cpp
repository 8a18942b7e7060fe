// Histogram op for Scanner on MI355X.
//
// Drop-in for the reference's kernels
//   HistogramKernelCPU  /root/reference/scannertools/scannertools_cpp/imgproc/histogram_kernel_cpu.cpp:11-57
//   HistogramKernelGPU  .../histogram_kernel_gpu.cpp:12-82 (OpenCV-CUDA wrapper, --build-cuda only)
// Same op declaration, same element format (3 x BINS int32, channel-major, one element per
// row), same registration shape (.device(GPU).batch().num_devices(1)); the per-frame
// cv::calcHist / cvc::histEven calls are replaced by ONE st_hist_u8c3_batch() call per
// execute(), i.e. one HIP launch for the whole batch.
//
// Extension over the reference (which has no op arguments): if KernelConfig::args holds a
// 4-byte little-endian int32 it is taken as the bin count (1..256); the default is the
// reference's BINS = 16.
#include <cstring>

#include "scanner/api/kernel.h"
#include "scanner/api/op.h"
#include "scanner/util/hip.h"
#include "scanner/util/memory.h"
#include "scannertools_hip.h"
#include "stage.h"

namespace scanner {
namespace {
const i32 BINS = 16;  // histogram_kernel_cpu.cpp:8
}

class HistogramKernelHIP : public BatchedKernel, public VideoKernel {
 public:
  HistogramKernelHIP(const KernelConfig& config)
    : BatchedKernel(config), device_(config.devices[0]), bins_(BINS) {
    if (config.args.size() == sizeof(i32)) memcpy(&bins_, config.args.data(), sizeof(i32));
    if (device_.type != DeviceType::GPU) {
      RESULT_ERROR(&valid_, "HistogramKernelHIP runs on DeviceType::GPU only");
    } else if (bins_ < 1 || bins_ > 256) {
      RESULT_ERROR(&valid_, "Histogram bins must be in [1, 256], got %d", bins_);
    } else {
      int st = st_ctx_create(device_.id, &ctx_);
      if (st != ST_OK) RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s", device_.id, st_status_string(st));
    }
  }

  ~HistogramKernelHIP() {
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
        << "Histogram expects U8 frames with 3 channels";

    size_t hist_size = bins_ * 3 * sizeof(i32);
    // one device block for the whole batch, one reference per output element
    u8* output_block = new_block_buffer(device_, hist_size * input_count, input_count);

    frames_.resize(input_count);
    for (i32 i = 0; i < input_count; ++i) {
      const Frame* f = frame_col[i].as_const_frame();
      LOG_IF(FATAL, f->as_frame_info() != frame_info_) << "Histogram: frame " << i << " changes shape inside a batch";
      frames_[i] = f->data;
    }
    int st = st_hist_u8c3_batch(ctx_, frames_.data(), input_count, frame_info_.height(), frame_info_.width(), bins_,
                                (int32_t*)output_block);
    LOG_IF(FATAL, st != ST_OK) << "st_hist_u8c3_batch: " << st_ctx_last_error(ctx_);
    st = st_ctx_sync(ctx_);  // the engine may read the elements from another stream
    LOG_IF(FATAL, st != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(ctx_);

    for (i32 i = 0; i < input_count; ++i) insert_element(output_columns[0], output_block + i * hist_size, hist_size);
  }

 private:
  DeviceHandle device_;
  i32 bins_;
  Result valid_;
  st_ctx* ctx_ = nullptr;
  std::vector<const uint8_t*> frames_;
};

// Same op for graphs that keep the reference's default device (CPU): host frames in, host
// elements out (histogram_kernel_cpu.cpp:16-46), computed on the GPU through a staging buffer.
class HistogramKernelHIPStaged : public BatchedKernel, public VideoKernel {
 public:
  HistogramKernelHIPStaged(const KernelConfig& config)
    : BatchedKernel(config), device_(config.devices[0]), bins_(BINS), gpu_(staging_device_id()), stage_(gpu_) {
    if (config.args.size() == sizeof(i32)) memcpy(&bins_, config.args.data(), sizeof(i32));
    if (bins_ < 1 || bins_ > 256) {
      RESULT_ERROR(&valid_, "Histogram bins must be in [1, 256], got %d", bins_);
    } else {
      int st = st_ctx_create(gpu_, &ctx_);
      if (st != ST_OK) RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s (no CPU fallback exists)", gpu_, st_status_string(st));
    }
  }
  ~HistogramKernelHIPStaged() {
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
        << "Histogram expects U8 frames with 3 channels";
    size_t hist_size = bins_ * 3 * sizeof(i32);
    size_t frame_bytes = frame_info_.size(), stride = DeviceStage::align(frame_bytes);
    u8* dev = stage_.reserve(stride * input_count + hist_size * input_count);
    for (i32 i = 0; i < input_count; ++i) stage_.upload(dev + stride * i, frame_col[i].as_const_frame()->data, frame_bytes);
    u8* dev_out = dev + stride * input_count;
    int st = st_hist_u8c3_strided(ctx_, dev, stride, input_count, frame_info_.height(), frame_info_.width(), bins_,
                                  (int32_t*)dev_out);
    LOG_IF(FATAL, st != ST_OK) << "st_hist_u8c3_strided: " << st_ctx_last_error(ctx_);
    LOG_IF(FATAL, st_ctx_sync(ctx_) != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(ctx_);
    u8* output_block = new_block_buffer_size(device_, hist_size, input_count);
    stage_.download(output_block, dev_out, hist_size * input_count);
    for (i32 i = 0; i < input_count; ++i) insert_element(output_columns[0], output_block + i * hist_size, hist_size);
  }

 private:
  DeviceHandle device_;
  i32 bins_;
  int gpu_;
  DeviceStage stage_;
  Result valid_;
  st_ctx* ctx_ = nullptr;
};

REGISTER_OP(Histogram).frame_input("frame").output("histogram", ColumnType::Bytes, "Histogram");

REGISTER_KERNEL(Histogram, HistogramKernelHIPStaged)
    .device(DeviceType::CPU)
    .batch()
    .num_devices(1);

REGISTER_KERNEL(Histogram, HistogramKernelHIP)
    .device(DeviceType::GPU)
    .batch()
    .num_devices(1);
}
