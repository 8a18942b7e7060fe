// Host <-> device staging for kernels registered on DeviceType::CPU: Scanner hands CPU kernels
// frames in host memory (the reference's default device, e.g. sc.ops.Histogram(frame=...) in
// tests/test_all.py:225).  The arithmetic still runs on the MI355X: frames go up through one
// grow-only device buffer, results come back into the Scanner-allocated host outputs.
#pragma once
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

// GPU that backs CPU-registered kernels: SCANNERTOOLS_HIP_DEVICE (default 0)
inline int staging_device_id() {
  const char* e = getenv("SCANNERTOOLS_HIP_DEVICE");
  return e ? atoi(e) : 0;
}

}  // namespace scanner
