// Host <-> device staging for kernels registered on DeviceType::CPU: Scanner hands CPU kernels
// frames in host memory (the reference's default device, e.g. sc.ops.Histogram(frame=...) in
// tests/test_all.py:225).  The arithmetic still runs on the MI355X: frames go up through one
// grow-only device buffer, results come back into the Scanner-allocated host outputs.
#pragma once
#include <algorithm>
#include <cstring>

#include "scanner/util/hip.h"

namespace scanner {

class DeviceStage {
 public:
  explicit DeviceStage(int device_id) : device_id_(device_id) {}
  ~DeviceStage() {
    if (buf_) { (void)hipSetDevice(device_id_); (void)hipFree(buf_); }
  }
  // at least `bytes` of device memory, 256-byte aligned
  u8* reserve(size_t bytes) {
    if (bytes > cap_) {
      HIP_CHECK(hipSetDevice(device_id_));
      if (buf_) HIP_CHECK(hipFree(buf_));
      buf_ = nullptr;
      HIP_CHECK(hipMalloc((void**)&buf_, bytes));
      cap_ = bytes;
    }
    return buf_;
  }
  void upload(u8* dst_dev, const u8* src_host, size_t n) { HIP_CHECK(hipMemcpy(dst_dev, src_host, n, hipMemcpyHostToDevice)); }
  void download(u8* dst_host, const u8* src_dev, size_t n) { HIP_CHECK(hipMemcpy(dst_host, src_dev, n, hipMemcpyDeviceToHost)); }
  static size_t align(size_t v) { return (v + 255) / 256 * 256; }

 private:
  int device_id_;
  u8* buf_ = nullptr;
  size_t cap_ = 0;
};

// Pipelined host -> device upload for kernels registered on DeviceType::CPU whose work per frame is
// far shorter than the frame's trip over PCIe (Histogram: 0.3 us of kernel per 110 us of upload at
// 1080p).  A batch is cut into sub-batches that alternate between two slots; per slot a device
// buffer and -- only if a source frame turns out to be pageable -- a page-locked bounce buffer.
//   copy stream   : hipMemcpyAsync of sub-batch k+1 (straight from Scanner's buffer when it is
//                   page-locked -- Scanner allocates its CPU frame pools with cudaMallocHost when
//                   GPUs are present --, through the slot's bounce buffer otherwise)
//   compute stream: waits for the slot's upload event, runs the caller's kernel on sub-batch k
// so uploads run back to back and the kernels hide under them.  Counterpart of the reference GPU
// kernel receiving device frames directly (frame_to_gpu_mat, histogram_kernel_gpu.cpp:48).
class UploadPipeline {
 public:
  explicit UploadPipeline(int device_id) : device_id_(device_id) {}
  ~UploadPipeline() {
    (void)hipSetDevice(device_id_);
    for (int s = 0; s < 2; ++s) {
      if (dev_[s]) (void)hipFree(dev_[s]);
      if (pin_[s]) (void)hipHostFree(pin_[s]);
      if (up_[s]) (void)hipEventDestroy(up_[s]);
      if (done_[s]) (void)hipEventDestroy(done_[s]);
    }
    if (copy_) (void)hipStreamDestroy(copy_);
    if (comp_) (void)hipStreamDestroy(comp_);
  }
  // false if the streams / events cannot be created
  bool init() {
    if (hipSetDevice(device_id_) != hipSuccess) return false;
    if (hipStreamCreateWithFlags(&copy_, hipStreamNonBlocking) != hipSuccess) return false;
    if (hipStreamCreateWithFlags(&comp_, hipStreamNonBlocking) != hipSuccess) return false;
    for (int s = 0; s < 2; ++s)
      if (hipEventCreateWithFlags(&up_[s], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&done_[s], hipEventDisableTiming) != hipSuccess) return false;
    return true;
  }
  hipStream_t compute_stream() const { return comp_; }
  static bool is_page_locked(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
  }
  // Runs `launch(dev_base, first, count)` on the compute stream for every sub-batch of `sub` frames,
  // dev_base + i * stride holding frame first + i.  src(i) is the host pointer of frame i.
  template <typename Src, typename Launch>
  void run(i32 n, i32 sub, size_t frame_bytes, size_t stride, Src src, Launch launch) {
    HIP_CHECK(hipSetDevice(device_id_));
    if (sub < 1) sub = 1;
    const size_t need = stride * (size_t)sub;
    for (int s = 0; s < 2; ++s)
      if (need > cap_[s]) {
        if (busy_[s]) { HIP_CHECK(hipEventSynchronize(done_[s])); busy_[s] = false; }
        if (dev_[s]) HIP_CHECK(hipFree(dev_[s]));
        dev_[s] = nullptr;
        HIP_CHECK(hipMalloc((void**)&dev_[s], need));
        cap_[s] = need;
      }
    i32 k = 0;
    for (i32 first = 0; first < n; first += sub, ++k) {
      const int s = k & 1;
      const i32 nb = std::min(sub, n - first);
      // the slot's previous kernel (and therefore its upload) must be done before its buffers are reused
      if (busy_[s]) HIP_CHECK(hipEventSynchronize(done_[s]));
      for (i32 i = 0; i < nb; ++i) {
        const u8* p = src(first + i);
        if (!is_page_locked(p)) {
          if (need > pin_cap_[s]) {
            if (pin_[s]) HIP_CHECK(hipHostFree(pin_[s]));
            pin_[s] = nullptr;
            HIP_CHECK(hipHostMalloc((void**)&pin_[s], need, hipHostMallocDefault));
            pin_cap_[s] = need;
          }
          memcpy(pin_[s] + stride * i, p, frame_bytes);
          p = pin_[s] + stride * i;
        }
        HIP_CHECK(hipMemcpyAsync(dev_[s] + stride * i, p, frame_bytes, hipMemcpyHostToDevice, copy_));
      }
      HIP_CHECK(hipEventRecord(up_[s], copy_));
      HIP_CHECK(hipStreamWaitEvent(comp_, up_[s], 0));
      launch(dev_[s], first, nb);
      HIP_CHECK(hipEventRecord(done_[s], comp_));
      busy_[s] = true;
    }
  }
  // all uploads and kernels of the last run() have finished
  void drain() {
    HIP_CHECK(hipStreamSynchronize(comp_));
    busy_[0] = busy_[1] = false;
  }

 private:
  int device_id_;
  hipStream_t copy_ = nullptr, comp_ = nullptr;
  hipEvent_t up_[2] = {nullptr, nullptr}, done_[2] = {nullptr, nullptr};
  u8* dev_[2] = {nullptr, nullptr};
  u8* pin_[2] = {nullptr, nullptr};
  size_t cap_[2] = {0, 0}, pin_cap_[2] = {0, 0};
  bool busy_[2] = {false, false};
};

// GPU that backs CPU-registered kernels: SCANNERTOOLS_HIP_DEVICE (default 0)
inline int staging_device_id() {
  const char* e = getenv("SCANNERTOOLS_HIP_DEVICE");
  return e ? atoi(e) : 0;
}

}  // namespace scanner
