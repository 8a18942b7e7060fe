// Kernel API subset (scanner/api/kernel.h): element containers, kernel base classes and the
// REGISTER_KERNEL builder.  Reference uses: BatchedKernel (histogram_kernel_cpu.cpp:11-17),
// StenciledKernel + VideoKernel (optical_flow_kernel_cpu.cpp:10,19-30), StenciledBatchedKernel
// (optical_flow_kernel_gpu.cpp:12,45-46), Kernel (blur_kernel_cpu.cpp:25,51-52);
// REGISTER_KERNEL(...).device().batch().num_devices() (histogram_kernel_cpu.cpp:54-57).
#pragma once
#include "scanner/util/profiler.h"
#include <cstring>
#include <functional>
#include <map>
#include <memory>

#include "scanner/api/frame.h"

namespace scanner {

struct Element {
  Element() = default;
  Element(u8* b, size_t s) : buffer(b), size(s), is_frame(false) {}
  explicit Element(Frame* f) : buffer((u8*)f), size(sizeof(Frame)), is_frame(true) {}
  const Frame* as_const_frame() const { return reinterpret_cast<const Frame*>(buffer); }
  Frame* as_frame() { return reinterpret_cast<Frame*>(buffer); }
  bool is_null() const { return buffer == nullptr; }
  u8* buffer = nullptr;
  size_t size = 0;
  bool is_frame = false;
  i64 index = 0;  // row index in the stream
};

using Elements = std::vector<Element>;
using BatchedElements = std::vector<Elements>;                        // [column][row]
using StenciledElements = std::vector<Elements>;                      // [column][stencil]
using StenciledBatchedElements = std::vector<std::vector<Elements>>;  // [column][row][stencil]

inline size_t num_rows(const Elements& column) { return column.size(); }

inline void insert_element(Elements& column, u8* buffer, size_t size) { column.emplace_back(buffer, size); }
inline void insert_frame(Elements& column, Frame* frame) { column.emplace_back(frame); }
inline void insert_element(Element& element, u8* buffer, size_t size) { element = Element(buffer, size); }
inline void insert_frame(Element& element, Frame* frame) { element = Element(frame); }

struct KernelConfig {
  std::vector<DeviceHandle> devices;
  std::vector<std::string> input_columns;
  std::vector<std::string> output_columns;
  std::vector<u8> args;
  i32 node_id = 0;
};

// scanner/util/profiler.h: Profiler::add_interval + now(), what the reference's kernels record their work under

class BaseKernel {
 public:
  explicit BaseKernel(const KernelConfig& config) : config_(config) {}
  virtual ~BaseKernel() {}
  virtual void validate(Result* result) { result->set_success(true); }
  virtual void fetch_resources(Result* result) { result->set_success(true); }
  virtual void setup_with_resources(Result* result) { result->set_success(true); }
  virtual void new_stream(const std::vector<u8>& /*args*/) {}
  virtual void reset() {}
  void set_profiler(Profiler* p) { profiler_ = p; }
  const KernelConfig& config() const { return config_; }

 protected:
  KernelConfig config_;
  Profiler* profiler_ = nullptr;
};

class StenciledBatchedKernel : public BaseKernel {
 public:
  explicit StenciledBatchedKernel(const KernelConfig& c) : BaseKernel(c) {}
  virtual void execute(const StenciledBatchedElements& input_columns, BatchedElements& output_columns) = 0;
};

class BatchedKernel : public BaseKernel {
 public:
  explicit BatchedKernel(const KernelConfig& c) : BaseKernel(c) {}
  virtual void execute(const BatchedElements& input_columns, BatchedElements& output_columns) = 0;
};

class StenciledKernel : public BaseKernel {
 public:
  explicit StenciledKernel(const KernelConfig& c) : BaseKernel(c) {}
  virtual void execute(const StenciledElements& input_columns, Elements& output_columns) = 0;
};

class Kernel : public BaseKernel {
 public:
  explicit Kernel(const KernelConfig& c) : BaseKernel(c) {}
  virtual void execute(const Elements& input_columns, Elements& output_columns) = 0;
};

// Mixin for kernels that care about the frame geometry: check_frame() calls new_frame_info()
// whenever the incoming FrameInfo changes (optical_flow_kernel_cpu.cpp:19-30).
class VideoKernel {
 public:
  virtual ~VideoKernel() {}

 protected:
  void check_frame(const DeviceHandle& /*device*/, const Element& element) {
    const Frame* f = element.as_const_frame();
    FrameInfo info = f->as_frame_info();
    if (!have_info_ || info != frame_info_) {
      frame_info_ = info;
      have_info_ = true;
      new_frame_info();
    }
  }
  // the FrameInfo travels as a bytes element (InfoFromFrame op, misc/info_from_frame_kernel.cpp:17-27):
  // the raw struct, on the kernel's device (host in every use of the reference: cpm2_output_kernel_cpu.cpp:147)
  void check_frame_info(const DeviceHandle& /*device*/, const Element& element) {
    FrameInfo info;
    if (element.size >= sizeof(FrameInfo)) memcpy((void*)&info, element.buffer, sizeof(FrameInfo));
    if (!have_info_ || info != frame_info_) {
      frame_info_ = info;
      have_info_ = true;
      new_frame_info();
    }
  }
  virtual void new_frame_info() {}
  FrameInfo frame_info_;

 private:
  bool have_info_ = false;
};

// ---- registration ---------------------------------------------------------------------------
using KernelConstructor = std::function<BaseKernel*(const KernelConfig&)>;
enum class KernelKind { Plain, Batched, Stenciled, StenciledBatched };

struct KernelRegistration {
  std::string op_name;
  KernelConstructor constructor;
  KernelKind kind = KernelKind::Plain;
  DeviceType device_type = DeviceType::CPU;
  i32 num_devices = 1;
  bool can_batch = false;
  i32 preferred_batch = 1;
};

class KernelBuilder {
 public:
  KernelBuilder(const std::string& op, KernelConstructor ctor, KernelKind kind);
  ~KernelBuilder();  // commits the registration
  KernelBuilder& device(DeviceType t) { reg_.device_type = t; return *this; }
  KernelBuilder& num_devices(i32 n) { reg_.num_devices = n; return *this; }
  KernelBuilder& batch(i32 preferred = 1) { reg_.can_batch = true; reg_.preferred_batch = preferred; return *this; }

 private:
  KernelRegistration reg_;
};

template <typename K> struct kernel_kind_of {
  static constexpr KernelKind value =
      std::is_base_of<StenciledBatchedKernel, K>::value ? KernelKind::StenciledBatched
      : std::is_base_of<BatchedKernel, K>::value        ? KernelKind::Batched
      : std::is_base_of<StenciledKernel, K>::value      ? KernelKind::Stenciled
                                                        : KernelKind::Plain;
};

#define ST_SHIM_CAT2(a, b) a##b
#define ST_SHIM_CAT(a, b) ST_SHIM_CAT2(a, b)
#define REGISTER_KERNEL(name__, kernel__)                                                          \
  static ::scanner::KernelBuilder ST_SHIM_CAT(kernel_registration_, __COUNTER__) __attribute__((unused)) = \
      ::scanner::KernelBuilder(#name__,                                                            \
                               [](const ::scanner::KernelConfig& c) -> ::scanner::BaseKernel* { return new kernel__(c); }, \
                               ::scanner::kernel_kind_of<kernel__>::value)

}  // namespace scanner
