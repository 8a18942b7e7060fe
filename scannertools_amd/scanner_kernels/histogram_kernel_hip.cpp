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
// Extension over the reference (which has no op arguments): the op declares
// protobuf_name("HistogramArgs") -- message HistogramArgs { int32 bins = 1; }
// (scannertools_imgproc_amd.proto beside this file) -- so that sc.ops.Histogram(frame=..., bins=256)
// reaches the kernel through Scanner's ordinary argument path.  Without arguments (every existing
// graph) the bin count is the reference's BINS = 16.
#include <cstring>

#include "scanner/api/kernel.h"
#include "scanner/api/op.h"
#include "scanner/util/hip.h"
#include "scanner/util/memory.h"
#include "proto_lite.h"
#include "scannertools_hip.h"
#include "stage.h"

namespace scanner {
namespace {
const i32 BINS = 16;  // histogram_kernel_cpu.cpp:8

// HistogramArgs { int32 bins = 1; }; empty args = the reference's op (16 bins).  false: malformed.
bool parse_histogram_args(const std::vector<u8>& args, i32* bins) {
  *bins = BINS;
  if (args.empty()) return true;
  std::vector<proto_lite::Field> fields;
  if (!proto_lite::parse(args.data(), args.size(), &fields)) return false;
  for (auto& f : fields)
    if (f.number == 1 && f.wire == 0) *bins = (i32)f.value;
  return true;
}
}

class HistogramKernelHIP : public BatchedKernel, public VideoKernel {
 public:
  HistogramKernelHIP(const KernelConfig& config)
    : BatchedKernel(config), device_(config.devices[0]), bins_(BINS) {
    if (!parse_histogram_args(config.args, &bins_)) {
      RESULT_ERROR(&valid_, "Could not parse HistogramArgs");
    } else if (device_.type != DeviceType::GPU) {
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
// elements out (histogram_kernel_cpu.cpp:16-46), computed on the GPU.  The kernel needs 0.3 us per
// 1080p frame and the upload 110 us, so the work is organising the uploads: sub-batches alternate
// between two device slots, uploads run back to back on a copy stream (straight out of Scanner's
// page-locked frame buffers; through a page-locked bounce ring when a buffer is pageable) while the
// compute stream histograms the sub-batch that has just arrived (stage.h: UploadPipeline).
class HistogramKernelHIPStaged : public BatchedKernel, public VideoKernel {
 public:
  HistogramKernelHIPStaged(const KernelConfig& config)
    : BatchedKernel(config), device_(config.devices[0]), bins_(BINS), gpu_(staging_device_id()), pipe_(gpu_), out_stage_(gpu_) {
    const char* e = getenv("SCANNERTOOLS_HIST_SUBBATCH");
    sub_ = e ? atoi(e) : 8;
    if (sub_ < 1) sub_ = 1;
    if (!parse_histogram_args(config.args, &bins_)) {
      RESULT_ERROR(&valid_, "Could not parse HistogramArgs");
    } else if (bins_ < 1 || bins_ > 256) {
      RESULT_ERROR(&valid_, "Histogram bins must be in [1, 256], got %d", bins_);
    } else {
      int st = st_ctx_create(gpu_, &ctx_);
      if (st != ST_OK) {
        RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s (no CPU fallback exists)", gpu_, st_status_string(st));
      } else if (!pipe_.init() || st_ctx_set_stream(ctx_, pipe_.compute_stream()) != ST_OK) {
        RESULT_ERROR(&valid_, "cannot create the upload pipeline on device %d", gpu_);
      }
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
    for (i32 i = 0; i < input_count; ++i)
      LOG_IF(FATAL, frame_col[i].as_const_frame()->as_frame_info() != frame_info_)
          << "Histogram: frame " << i << " changes shape inside a batch";
    const size_t hist_size = bins_ * 3 * sizeof(i32);
    const size_t frame_bytes = frame_info_.size(), stride = DeviceStage::align(frame_bytes);
    const i32 h = frame_info_.height(), w = frame_info_.width();
    u8* dev_out = out_stage_.reserve(hist_size * input_count);
    pipe_.run(input_count, sub_, frame_bytes, stride,
              [&](i32 i) { return (const u8*)frame_col[i].as_const_frame()->data; },
              [&](u8* dev, i32 first, i32 nb) {
                int st = st_hist_u8c3_strided(ctx_, dev, stride, nb, h, w, bins_, (int32_t*)(dev_out + hist_size * first));
                LOG_IF(FATAL, st != ST_OK) << "st_hist_u8c3_strided: " << st_ctx_last_error(ctx_);
              });
    u8* output_block = new_block_buffer_size(device_, hist_size, input_count);
    HIP_CHECK(hipMemcpyAsync(output_block, dev_out, hist_size * input_count, hipMemcpyDeviceToHost, pipe_.compute_stream()));
    pipe_.drain();
    for (i32 i = 0; i < input_count; ++i) insert_element(output_columns[0], output_block + i * hist_size, hist_size);
  }

 private:
  DeviceHandle device_;
  i32 bins_;
  int gpu_;
  int sub_ = 8;
  UploadPipeline pipe_;
  DeviceStage out_stage_;
  Result valid_;
  st_ctx* ctx_ = nullptr;
};

REGISTER_OP(Histogram).frame_input("frame").output("histogram", ColumnType::Bytes, "Histogram").protobuf_name("HistogramArgs");

REGISTER_KERNEL(Histogram, HistogramKernelHIPStaged)
    .device(DeviceType::CPU)
    .batch()
    .num_devices(1);

REGISTER_KERNEL(Histogram, HistogramKernelHIP)
    .device(DeviceType::GPU)
    .batch()
    .num_devices(1);
}
