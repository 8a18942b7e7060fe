// FlowHistogram op for Scanner on MI355X.
//
// Drop-in for the reference's kernel
//   FlowHistogramKernelCPU  /root/reference/scannertools/scannertools/old/cpp_ops/flow_histogram_kernel_cpu.cpp:12-67
// Same op declaration (frame_input("flow") -> output("histogram")), same element format
// (2 x 64 int32: magnitude histogram over [0,64) then angle histogram over [0,360) degrees, one
// 512-byte element per row) and the same registration shape (.batch().num_devices(1)).  The
// per-frame cv::split / cv::cartToPolar / cv::calcHist x2 calls are replaced by ONE
// st_flow_hist_batch() call per execute().  The reference registers the op on DeviceType::CPU
// only; here the CPU registration stages host flow frames through the GPU, and a
// DeviceType::GPU registration consumes OpticalFlow's device output without a round trip
// (the legacy pipeline Resize -> OpticalFlow -> FlowHistogram, old/histograms.py:63-78).
#include "scanner/api/kernel.h"
#include "scanner/api/op.h"
#include "scanner/util/hip.h"
#include "scanner/util/memory.h"
#include "scannertools_hip.h"
#include "stage.h"

namespace scanner {
namespace {
const i32 BINS = 64;  // flow_histogram_kernel_cpu.cpp:9
}

class FlowHistogramKernelHIP : public BatchedKernel, public VideoKernel {
 public:
  FlowHistogramKernelHIP(const KernelConfig& config) : BatchedKernel(config), device_(config.devices[0]) {
    if (device_.type != DeviceType::GPU) {
      RESULT_ERROR(&valid_, "FlowHistogramKernelHIP runs on DeviceType::GPU only");
    } else {
      int st = st_ctx_create(device_.id, &ctx_);
      if (st != ST_OK) RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s", device_.id, st_status_string(st));
    }
  }
  ~FlowHistogramKernelHIP() {
    if (ctx_) st_ctx_destroy(ctx_);
  }
  void validate(Result* result) override {
    result->set_msg(valid_.msg());
    result->set_success(valid_.success());
  }

  void execute(const BatchedElements& input_columns, BatchedElements& output_columns) override {
    auto& flow_col = input_columns[0];
    i32 input_count = (i32)num_rows(flow_col);
    if (input_count == 0) return;
    check_frame(device_, flow_col[0]);
    LOG_IF(FATAL, frame_info_.channels() != 2 || frame_info_.type != FrameType::F32)
        << "FlowHistogram expects F32 frames with 2 channels";
    size_t hist_size = BINS * 2 * sizeof(i32);
    u8* output_block = new_block_buffer(device_, hist_size * input_count, input_count);
    flows_.resize(input_count);
    for (i32 i = 0; i < input_count; ++i) {
      const Frame* f = flow_col[i].as_const_frame();
      LOG_IF(FATAL, f->as_frame_info() != frame_info_) << "FlowHistogram: frame " << i << " changes shape inside a batch";
      flows_[i] = (const float*)f->data;
    }
    int st = st_flow_hist_batch(ctx_, flows_.data(), input_count, frame_info_.height(), frame_info_.width(),
                                (int32_t*)output_block);
    LOG_IF(FATAL, st != ST_OK) << "st_flow_hist_batch: " << st_ctx_last_error(ctx_);
    st = st_ctx_sync(ctx_);  // the engine may read the elements from another stream
    LOG_IF(FATAL, st != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(ctx_);
    for (i32 i = 0; i < input_count; ++i) insert_element(output_columns[0], output_block + i * hist_size, hist_size);
  }

 private:
  DeviceHandle device_;
  Result valid_;
  st_ctx* ctx_ = nullptr;
  std::vector<const float*> flows_;
};

// The reference's registration: host flow frames in, host elements out
// (flow_histogram_kernel_cpu.cpp:18-57), computed on the GPU through a staging buffer.
class FlowHistogramKernelHIPStaged : public BatchedKernel, public VideoKernel {
 public:
  FlowHistogramKernelHIPStaged(const KernelConfig& config)
    : BatchedKernel(config), device_(config.devices[0]), gpu_(staging_device_id()), stage_(gpu_) {
    int st = st_ctx_create(gpu_, &ctx_);
    if (st != ST_OK) RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s (no CPU fallback exists)", gpu_, st_status_string(st));
  }
  ~FlowHistogramKernelHIPStaged() {
    if (ctx_) st_ctx_destroy(ctx_);
  }
  void validate(Result* result) override {
    result->set_msg(valid_.msg());
    result->set_success(valid_.success());
  }

  void execute(const BatchedElements& input_columns, BatchedElements& output_columns) override {
    auto& flow_col = input_columns[0];
    i32 input_count = (i32)num_rows(flow_col);
    if (input_count == 0) return;
    check_frame(device_, flow_col[0]);
    LOG_IF(FATAL, frame_info_.channels() != 2 || frame_info_.type != FrameType::F32)
        << "FlowHistogram expects F32 frames with 2 channels";
    size_t hist_size = BINS * 2 * sizeof(i32);
    size_t frame_bytes = frame_info_.size(), stride = DeviceStage::align(frame_bytes);
    u8* dev = stage_.reserve(stride * input_count + hist_size * input_count);
    for (i32 i = 0; i < input_count; ++i) stage_.upload(dev + stride * i, flow_col[i].as_const_frame()->data, frame_bytes);
    u8* dev_out = dev + stride * input_count;
    int st = st_flow_hist_strided(ctx_, (const float*)dev, stride, input_count, frame_info_.height(), frame_info_.width(),
                                  (int32_t*)dev_out);
    LOG_IF(FATAL, st != ST_OK) << "st_flow_hist_strided: " << st_ctx_last_error(ctx_);
    LOG_IF(FATAL, st_ctx_sync(ctx_) != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(ctx_);
    u8* output_block = new_block_buffer_size(device_, hist_size, input_count);
    stage_.download(output_block, dev_out, hist_size * input_count);
    for (i32 i = 0; i < input_count; ++i) insert_element(output_columns[0], output_block + i * hist_size, hist_size);
  }

 private:
  DeviceHandle device_;
  int gpu_;
  DeviceStage stage_;
  Result valid_;
  st_ctx* ctx_ = nullptr;
};

REGISTER_OP(FlowHistogram).frame_input("flow").output("histogram");

REGISTER_KERNEL(FlowHistogram, FlowHistogramKernelHIPStaged)
    .device(DeviceType::CPU)
    .batch()
    .num_devices(1);

REGISTER_KERNEL(FlowHistogram, FlowHistogramKernelHIP)
    .device(DeviceType::GPU)
    .batch()
    .num_devices(1);
}
