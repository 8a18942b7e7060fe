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
#include <algorithm>
#include <memory>
#include <string>
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
// frames out (optical_flow_kernel_cpu.cpp:27-43), computed on the GPU.  Registered batched (the
// reference's CPU kernel is not) so that a `batch=` on the op amortises the frames shared by
// consecutive windows; batch 1 reproduces the reference's calling pattern.
//
// The batch is cut into sub-batches that alternate between two lanes (stream + context + device
// staging buffer each): while lane A computes sub-batch i and copies its flow fields back
// (asynchronously, into the Scanner-allocated outputs), the host thread uploads the frames of
// sub-batch i+1 for lane B.  A 1080p flow frame is 16.6 MB going back for 6.2 MB coming in, so
// the path is PCIe-bound; overlapping the three stages is what this structure buys.
class OpticalFlowKernelHIPStaged : public StenciledBatchedKernel, public VideoKernel {
 public:
  OpticalFlowKernelHIPStaged(const KernelConfig& config)
    : StenciledBatchedKernel(config), device_(config.devices[0]), gpu_(staging_device_id()) {
    st_fb_params_default(&params_);
    const char* e = getenv("SCANNERTOOLS_FLOW_SUBBATCH");
    sub_fixed_ = e != nullptr;
    sub_ = e ? atoi(e) : 8;
    if (sub_ < 1) sub_ = 1;
    for (int l = 0; l < 2 && valid_.success(); ++l) {
      lanes_[l].stage.reset(new DeviceStage(gpu_));
      int st = st_ctx_create(gpu_, &lanes_[l].ctx);
      if (st != ST_OK) {
        RESULT_ERROR(&valid_, "st_ctx_create(%d) failed: %s (no CPU fallback exists)", gpu_, st_status_string(st));
        break;
      }
      if (hipStreamCreateWithFlags(&lanes_[l].stream, hipStreamNonBlocking) != hipSuccess ||
          st_ctx_set_stream(lanes_[l].ctx, lanes_[l].stream) != ST_OK ||
          hipEventCreateWithFlags(&lanes_[l].up_done, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&lanes_[l].comp_done, hipEventDisableTiming) != hipSuccess)
        RESULT_ERROR(&valid_, "cannot create a HIP stream on device %d", gpu_);
    }
    // SCANNERTOOLS_FLOW_COPIES=overlap: every lane copies on its own stream (uploads of one lane run beside the other's
    // copy-back); default "serial": ALL copies of both lanes on one stream, so that an upload and a copy-back are never in
    // flight together -- this host's device-to-host copies run at 57 GB/s alone and at 36-38 GB/s beside an upload, and
    // the flow fields going back are 2.7x the frames coming in
    const char* cm = getenv("SCANNERTOOLS_FLOW_COPIES");
    serial_copies_ = !(cm && std::string(cm) == "overlap");
    if (valid_.success() && hipStreamCreateWithFlags(&copy_, hipStreamNonBlocking) != hipSuccess)
      RESULT_ERROR(&valid_, "cannot create a HIP stream on device %d", gpu_);
  }
  ~OpticalFlowKernelHIPStaged() {
    if (copy_) { (void)hipStreamSynchronize(copy_); (void)hipStreamDestroy(copy_); }
    for (auto& l : lanes_) {
      if (l.up_done) (void)hipEventDestroy(l.up_done);
      if (l.comp_done) (void)hipEventDestroy(l.comp_done);
      if (l.ctx) st_ctx_destroy(l.ctx);
      if (l.stream) (void)hipStreamDestroy(l.stream);
    }
  }
  void validate(Result* result) override {
    result->set_msg(valid_.msg());
    result->set_success(valid_.success());
  }
  void new_frame_info() override {
    for (auto& l : lanes_) if (l.ctx) st_ctx_release_workspace(l.ctx);
  }

  void execute(const StenciledBatchedElements& input_columns, BatchedElements& output_columns) override {
    auto& frame_col = input_columns[0];
    i32 input_count = (i32)frame_col.size();
    if (input_count == 0) return;
    check_frame(device_, frame_col[0][0]);
    LOG_IF(FATAL, frame_info_.channels() != 3 || frame_info_.type != FrameType::U8)
        << "OpticalFlow expects U8 frames with 3 channels";
    for (i32 i = 0; i < input_count; ++i) {
      LOG_IF(FATAL, frame_col[i].size() != 2) << "OpticalFlow needs a 2-element stencil, got " << frame_col[i].size();
      for (i32 s = 0; s < 2; ++s)
        LOG_IF(FATAL, frame_col[i][s].as_const_frame()->as_frame_info() != frame_info_)
            << "OpticalFlow: frame shape changes inside a batch";
    }
    FrameInfo out_info(frame_info_.height(), frame_info_.width(), 2, FrameType::F32);
    const size_t frame_bytes = frame_info_.size(), fstride = DeviceStage::align(frame_bytes);
    const size_t out_bytes = out_info.size(), ostride = DeviceStage::align(out_bytes);
    std::vector<Frame*> output_frames = new_frames(device_, out_info, input_count);
    HIP_CHECK(hipSetDevice(gpu_));

    // Sub-batch size: SCANNERTOOLS_FLOW_SUBBATCH when set; otherwise by bytes -- about 64 MB of frames + flow fields per
    // sub-batch, at least 8 rows (1080p: 8 rows = 180 MB), so that small frames (the legacy pipeline's 426x240: 1.1 MB per
    // row) are not cut into pieces whose fixed costs (two synchronisations, a dozen launches, two copies) outweigh them:
    // 64 rows of 426x240 went from 7.0 k to the rate printed by bench.py's legacy_flow_hist record.  Even split.
    i32 sub = sub_;
    if (!sub_fixed_) {
      const size_t per_row = frame_bytes + out_bytes;
      sub = (i32)std::min<size_t>(64, std::max<size_t>(8, ((size_t)64 << 20) / std::max<size_t>(per_row, 1)));
      const i32 pieces = (input_count + sub - 1) / sub;
      sub = (input_count + pieces - 1) / pieces;
    }
    // copy-back of the sub-batch a lane computed last (serial mode: enqueued on the shared copy stream behind the NEXT
    // sub-batch's upload, so that uploads stay one sub-batch ahead of the compute)
    struct Pending { bool any = false; i32 r0 = 0, nb = 0; std::vector<float*> dev_outs; };
    Pending pending[2];
    auto copy_back = [&](int li) {
      Pending& P = pending[li];
      if (!P.any) return;
      HIP_CHECK(hipStreamWaitEvent(copy_, lanes_[li].comp_done, 0));
      for (i32 i = 0; i < P.nb;) {
        i32 j = i + 1;
        while (ostride == out_bytes && j < P.nb && output_frames[P.r0 + j]->data == output_frames[P.r0 + j - 1]->data + out_bytes) ++j;
        HIP_CHECK(hipMemcpyAsync(output_frames[P.r0 + i]->data, P.dev_outs[i], out_bytes * (size_t)(j - i), hipMemcpyDeviceToHost, copy_));
        i = j;
      }
      P.any = false;
    };
    int lane_idx = 0;
    for (i32 r0 = 0; r0 < input_count; r0 += sub, lane_idx ^= 1) {
      const i32 nb = std::min(sub, input_count - r0);
      Lane& L = lanes_[lane_idx];
      hipStream_t up_stream = serial_copies_ ? copy_ : L.stream;
      // the lane's previous sub-batch (compute + copy-back) must be done before its buffers are reused: own stream in
      // overlap mode; in serial mode its copy-back goes first on the copy stream, and this sub-batch's compute waits for
      // the upload behind it
      if (serial_copies_) copy_back(lane_idx);
      else LOG_IF(FATAL, st_ctx_sync(L.ctx) != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(L.ctx);
      // distinct frames of this sub-batch
      std::vector<const u8*> host_frames;
      std::vector<int32_t> pairs;
      std::unordered_map<const u8*, i32> slot;
      for (i32 i = r0; i < r0 + nb; ++i)
        for (i32 s = 0; s < 2; ++s) {
          const u8* d = frame_col[i][s].as_const_frame()->data;
          auto it = slot.find(d);
          if (it == slot.end()) {
            it = slot.emplace(d, (i32)host_frames.size()).first;
            host_frames.push_back(d);
          }
          pairs.push_back(it->second);
        }
      u8* dev = L.stage->reserve(fstride * host_frames.size() + ostride * nb);
      std::vector<const uint8_t*> dev_frames(host_frames.size());
      // One copy per RUN of frames that are adjacent in host memory (Scanner hands out a batch's frames from block
      // allocations): a PCIe copy of one 6 MB frame carries ~0.2 ms of fixed cost, a third of its duration.
      // (Pageable sources: the call returns when the data is on its way; the other lane keeps computing.)
      for (size_t i = 0; i < host_frames.size();) {
        size_t j = i + 1;
        while (fstride == frame_bytes && j < host_frames.size() && host_frames[j] == host_frames[j - 1] + frame_bytes) ++j;
        HIP_CHECK(hipMemcpyAsync(dev + fstride * i, host_frames[i], frame_bytes * (j - i), hipMemcpyHostToDevice, up_stream));
        for (; i < j; ++i) dev_frames[i] = dev + fstride * i;
      }
      if (serial_copies_) {
        HIP_CHECK(hipEventRecord(L.up_done, copy_));
        HIP_CHECK(hipStreamWaitEvent(L.stream, L.up_done, 0));
        copy_back(lane_idx ^ 1);  // the other lane's finished sub-batch comes back behind this upload
      }
      std::vector<float*> dev_outs(nb);
      for (i32 i = 0; i < nb; ++i) dev_outs[i] = (float*)(dev + fstride * host_frames.size() + ostride * i);
      int st = st_farneback_pairs(L.ctx, dev_frames.data(), (int)dev_frames.size(), pairs.data(), nb,
                                  frame_info_.height(), frame_info_.width(), &params_, dev_outs.data());
      LOG_IF(FATAL, st != ST_OK) << "st_farneback_pairs: " << st_ctx_last_error(L.ctx);
      if (serial_copies_) {
        HIP_CHECK(hipEventRecord(L.comp_done, L.stream));
        pending[lane_idx].any = true; pending[lane_idx].r0 = r0; pending[lane_idx].nb = nb; pending[lane_idx].dev_outs = dev_outs;
        continue;
      }
      // the output frames of a batch are one host block (new_frames) and the staged flows are adjacent on the device
      // whenever a flow field is a multiple of the staging alignment: then the whole sub-batch comes back in ONE copy
      // (16.6 MB copies reach 33 GB/s on this host, a 133 MB copy 57)
      for (i32 i = 0; i < nb;) {
        i32 j = i + 1;
        while (ostride == out_bytes && j < nb && output_frames[r0 + j]->data == output_frames[r0 + j - 1]->data + out_bytes) ++j;
        HIP_CHECK(hipMemcpyAsync(output_frames[r0 + i]->data, dev_outs[i], out_bytes * (size_t)(j - i), hipMemcpyDeviceToHost, L.stream));
        i = j;
      }
    }
    if (serial_copies_) {
      // the last two sub-batches, oldest first
      copy_back(lane_idx);
      copy_back(lane_idx ^ 1);
      HIP_CHECK(hipStreamSynchronize(copy_));
    }
    for (auto& l : lanes_) LOG_IF(FATAL, st_ctx_sync(l.ctx) != ST_OK) << "st_ctx_sync: " << st_ctx_last_error(l.ctx);
    for (i32 i = 0; i < input_count; ++i) insert_frame(output_columns[0], output_frames[i]);
  }

 private:
  struct Lane {
    st_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t up_done = nullptr, comp_done = nullptr;
    std::unique_ptr<DeviceStage> stage;
  };
  hipStream_t copy_ = nullptr;
  bool serial_copies_ = true;
  DeviceHandle device_;
  int gpu_;
  int sub_ = 8;
  bool sub_fixed_ = false;
  Lane lanes_[2];
  Result valid_;
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
