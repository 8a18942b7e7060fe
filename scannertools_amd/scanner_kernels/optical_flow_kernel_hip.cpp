// OpticalFlow op for Scanner on MI355X.
//
// Drop-in for the reference's kernels
//   OpticalFlowKernelCPU  /root/reference/scannertools/scannertools_cpp/imgproc/optical_flow_kernel_cpu.cpp:10-58
//   OpticalFlowKernelGPU  .../optical_flow_kernel_gpu.cpp:12-112 (OpenCV-CUDA wrapper, --build-cuda only)
// Same op declaration (frame in, frame out, stencil {0,1}) and the GPU wrapper's batched-stenciled
// registration.  Semantics follow the CPU kernel: output row i is the Farneback flow FROM stencil
// element 0 TO stencil element 1 of row i with create(3, 0.5, false, 15, 3, 5, 1.2, 0), after
// cv::cvtColor(COLOR_BGR2GRAY) of both frames.  Two defects of the reference GPU wrapper are
// not reproduced: its reversed (later, earlier) argument order (optical_flow_kernel_gpu.cpp:82-87)
// and its assumption that row i's second stencil element is row i+1's first (:53-57) -- every
// row's own window is honoured, and frames shared between windows are recognised by buffer
// address so that their pyramids are built once per execute().
#include <unordered_map>

#include "scanner/api/kernel.h"
#include "scanner/api/op.h"
#include "scanner/util/hip.h"
#include "scanner/util/memory.h"
#include "scannertools_hip.h"
#include "stage.h"

namespace scanner {

class OpticalFlowKernelHIP : public StenciledBatchedKernel, public VideoKernel {
 public:
  OpticalFlowKernelHIP(const KernelConfig& config)
    : StenciledBatchedKernel(config), device_(config.devices[0]) {
    st_fb_params_default(&params_);  // (3, 0.5, false, 15, 3, 5, 1.2, 0): optical_flow_kernel_cpu.cpp:16
    if (device_.type != DeviceType::GPU) {
      RESULT_ERROR(&valid_, "OpticalFlowKernelHIP runs on DeviceType::GPU only");
    } else {
      int st = st_ctx_create(device_.id, &ctx_);
      if (st != ST_OK) RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s", device_.id, st_status_string(st));
    }
  }

  ~OpticalFlowKernelHIP() {
    if (ctx_) st_ctx_destroy(ctx_);
  }

  void validate(Result* result) override {
    result->set_msg(valid_.msg());
    result->set_success(valid_.success());
  }

  void new_frame_info() override {
    // scratch is sized for the new geometry on the next call; drop the old one now
    if (ctx_) st_ctx_release_workspace(ctx_);
  }

  void reset() override {}

  void execute(const StenciledBatchedElements& input_columns, BatchedElements& output_columns) override {
    auto& frame_col = input_columns[0];
    i32 input_count = (i32)frame_col.size();
    if (input_count == 0) return;
    check_frame(device_, frame_col[0][0]);
    LOG_IF(FATAL, frame_info_.channels() != 3 || frame_info_.type != FrameType::U8)
        << "OpticalFlow expects U8 frames with 3 channels";

    // distinct frames of the batch (by buffer) and the (from, to) index pair of every row
    frames_.clear();
    pairs_.clear();
    std::unordered_map<const u8*, i32> slot;
    for (i32 i = 0; i < input_count; ++i) {
      LOG_IF(FATAL, frame_col[i].size() != 2) << "OpticalFlow needs a 2-element stencil, got " << frame_col[i].size();
      for (i32 s = 0; s < 2; ++s) {
        const Frame* f = frame_col[i][s].as_const_frame();
        LOG_IF(FATAL, f->as_frame_info() != frame_info_) << "OpticalFlow: frame shape changes inside a batch";
        auto it = slot.find(f->data);
        if (it == slot.end()) {
          it = slot.emplace(f->data, (i32)frames_.size()).first;
          frames_.push_back(f->data);
        }
        pairs_.push_back(it->second);
      }
    }

    FrameInfo out_frame_info(frame_info_.height(), frame_info_.width(), 2, FrameType::F32);
    std::vector<Frame*> output_frames = new_frames(device_, out_frame_info, input_count);
    outs_.resize(input_count);
    for (i32 i = 0; i < input_count; ++i) outs_[i] = (float*)output_frames[i]->data;

    int st = st_farneback_pairs(ctx_, frames_.data(), (int)frames_.size(), pairs_.data(), input_count,
                                frame_info_.height(), frame_info_.width(), &params_, outs_.data());
    LOG_IF(FATAL, st != ST_OK) << "st_farneback_pairs: " << st_ctx_last_error(ctx_);
    st = st_ctx_sync(ctx_);
    LOG_IF(FATAL, st != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(ctx_);

    for (i32 i = 0; i < input_count; ++i) insert_frame(output_columns[0], output_frames[i]);
  }

 private:
  DeviceHandle device_;
  Result valid_;
  st_ctx* ctx_ = nullptr;
  st_fb_params params_;
  std::vector<const uint8_t*> frames_;
  std::vector<int32_t> pairs_;
  std::vector<float*> outs_;
};

// Same op for graphs that keep the reference's default device (CPU): host frames in, host flow
// frames out (optical_flow_kernel_cpu.cpp:27-43), computed on the GPU through a staging buffer.
// Registered batched (the reference's CPU kernel is not) so that a `batch=` on the op amortises
// the shared frames of consecutive windows; batch 1 reproduces the reference's calling pattern.
class OpticalFlowKernelHIPStaged : public StenciledBatchedKernel, public VideoKernel {
 public:
  OpticalFlowKernelHIPStaged(const KernelConfig& config)
    : StenciledBatchedKernel(config), device_(config.devices[0]), gpu_(staging_device_id()), stage_(gpu_) {
    st_fb_params_default(&params_);
    int st = st_ctx_create(gpu_, &ctx_);
    if (st != ST_OK) RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s (no CPU fallback exists)", gpu_, st_status_string(st));
  }
  ~OpticalFlowKernelHIPStaged() {
    if (ctx_) st_ctx_destroy(ctx_);
  }
  void validate(Result* result) override {
    result->set_msg(valid_.msg());
    result->set_success(valid_.success());
  }
  void new_frame_info() override {
    if (ctx_) st_ctx_release_workspace(ctx_);
  }

  void execute(const StenciledBatchedElements& input_columns, BatchedElements& output_columns) override {
    auto& frame_col = input_columns[0];
    i32 input_count = (i32)frame_col.size();
    if (input_count == 0) return;
    check_frame(device_, frame_col[0][0]);
    LOG_IF(FATAL, frame_info_.channels() != 3 || frame_info_.type != FrameType::U8)
        << "OpticalFlow expects U8 frames with 3 channels";
    std::vector<const u8*> host_frames;
    std::vector<int32_t> pairs;
    std::unordered_map<const u8*, i32> slot;
    for (i32 i = 0; i < input_count; ++i) {
      LOG_IF(FATAL, frame_col[i].size() != 2) << "OpticalFlow needs a 2-element stencil, got " << frame_col[i].size();
      for (i32 s = 0; s < 2; ++s) {
        const Frame* f = frame_col[i][s].as_const_frame();
        LOG_IF(FATAL, f->as_frame_info() != frame_info_) << "OpticalFlow: frame shape changes inside a batch";
        auto it = slot.find(f->data);
        if (it == slot.end()) {
          it = slot.emplace(f->data, (i32)host_frames.size()).first;
          host_frames.push_back(f->data);
        }
        pairs.push_back(it->second);
      }
    }
    FrameInfo out_info(frame_info_.height(), frame_info_.width(), 2, FrameType::F32);
    size_t frame_bytes = frame_info_.size(), fstride = DeviceStage::align(frame_bytes), ostride = DeviceStage::align(out_info.size());
    u8* dev = stage_.reserve(fstride * host_frames.size() + ostride * input_count);
    std::vector<const uint8_t*> dev_frames(host_frames.size());
    for (size_t i = 0; i < host_frames.size(); ++i) {
      stage_.upload(dev + fstride * i, host_frames[i], frame_bytes);
      dev_frames[i] = dev + fstride * i;
    }
    std::vector<float*> dev_outs(input_count);
    for (i32 i = 0; i < input_count; ++i) dev_outs[i] = (float*)(dev + fstride * host_frames.size() + ostride * i);
    int st = st_farneback_pairs(ctx_, dev_frames.data(), (int)dev_frames.size(), pairs.data(), input_count,
                                frame_info_.height(), frame_info_.width(), &params_, dev_outs.data());
    LOG_IF(FATAL, st != ST_OK) << "st_farneback_pairs: " << st_ctx_last_error(ctx_);
    LOG_IF(FATAL, st_ctx_sync(ctx_) != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(ctx_);
    std::vector<Frame*> output_frames = new_frames(device_, out_info, input_count);
    for (i32 i = 0; i < input_count; ++i) {
      stage_.download(output_frames[i]->data, (const u8*)dev_outs[i], out_info.size());
      insert_frame(output_columns[0], output_frames[i]);
    }
  }

 private:
  DeviceHandle device_;
  int gpu_;
  DeviceStage stage_;
  Result valid_;
  st_ctx* ctx_ = nullptr;
  st_fb_params params_;
};

REGISTER_OP(OpticalFlow)
    .frame_input("frame")
    .frame_output("flow")
    .stencil({0, 1});

REGISTER_KERNEL(OpticalFlow, OpticalFlowKernelHIP)
    .device(DeviceType::GPU)
    .batch()
    .num_devices(1);

REGISTER_KERNEL(OpticalFlow, OpticalFlowKernelHIPStaged)
    .device(DeviceType::CPU)
    .batch()
    .num_devices(1);
}
