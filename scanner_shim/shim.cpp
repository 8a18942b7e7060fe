// Implementation of the Scanner API stand-in (see scanner/util/common.h) plus a mini engine that
// drives registered kernels the way Scanner's evaluator does for this path:
// one kernel instance, rows fed in batches, stencil windows assembled with edge clamping, one
// output element per input row, engine-owned output memory.  Exposed to Python through a small
// C API (stshim_*) so that the parity tests exercise Python front-end -> engine -> Scanner-style
// kernel class -> C ABI -> HIP.  None of this is needed inside a real Scanner deployment.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>

#include "scanner/api/kernel.h"
#include "scanner/api/op.h"

namespace scanner {

// ---- buffers ---------------------------------------------------------------------------------
namespace {
struct Block {
  DeviceHandle device;
  size_t size;
  i64 refs;
};
std::mutex g_mem_mutex;
std::map<uintptr_t, Block> g_blocks;  // base address -> block

std::map<uintptr_t, Block>::iterator find_block(const u8* p) {
  auto it = g_blocks.upper_bound((uintptr_t)p);
  if (it == g_blocks.begin()) return g_blocks.end();
  --it;
  if ((uintptr_t)p < it->first + std::max<size_t>(it->second.size, 1)) return it;
  return g_blocks.end();
}

// Host buffers come from a small pool of PINNED blocks (Scanner keeps pinned pools for the same
// reason): a D2H copy into freshly malloc'ed pageable memory is bound by first-touch page faults
// (~2 GB/s), not by PCIe.  Blocks are reused when a free one is at most 2x the request; the pool
// is capped so that it cannot grow without bound.  Without a GPU runtime plain malloc is used.
struct HostBlock { u8* p; size_t size; bool pinned; };
std::vector<HostBlock> g_host_pool;                 // free blocks
std::map<uintptr_t, HostBlock> g_host_live;         // blocks handed out
size_t g_host_pooled_bytes = 0;
constexpr size_t kHostPoolCap = (size_t)8 << 30;
constexpr size_t kHostPoolMin = (size_t)1 << 20;    // small buffers are not worth pinning

u8* host_alloc(size_t size) {
  {
    std::lock_guard<std::mutex> lk(g_mem_mutex);
    for (size_t i = 0; i < g_host_pool.size(); ++i) {
      if (g_host_pool[i].size >= size && g_host_pool[i].size <= 2 * size) {
        HostBlock b = g_host_pool[i];
        g_host_pool.erase(g_host_pool.begin() + i);
        g_host_pooled_bytes -= b.size;
        g_host_live[(uintptr_t)b.p] = b;
        return b.p;
      }
    }
  }
  HostBlock b{nullptr, size, false};
  if (size >= kHostPoolMin) {
    void* p = nullptr;
    if (hipHostMalloc(&p, size, hipHostMallocDefault) == hipSuccess) { b.p = (u8*)p; b.pinned = true; }
    else (void)hipGetLastError();
  }
  if (!b.p) b.p = (u8*)malloc(size);
  LOG_IF(FATAL, b.p == nullptr) << "host allocation failed";
  std::lock_guard<std::mutex> lk(g_mem_mutex);
  g_host_live[(uintptr_t)b.p] = b;
  return b.p;
}

void host_free(u8* p) {
  HostBlock b;
  {
    std::lock_guard<std::mutex> lk(g_mem_mutex);
    auto it = g_host_live.find((uintptr_t)p);
    LOG_IF(FATAL, it == g_host_live.end()) << "freeing unknown host block";
    b = it->second;
    g_host_live.erase(it);
    if (b.pinned && g_host_pooled_bytes + b.size <= kHostPoolCap) {
      g_host_pool.push_back(b);
      g_host_pooled_bytes += b.size;
      return;
    }
  }
  if (b.pinned) (void)hipHostFree(b.p); else free(b.p);
}

// Device buffers come from a pool too (Scanner's workers allocate element buffers from pooled blocks, not from the
// driver): a hipMalloc + hipFree pair per execute() costs tens of microseconds, which is a tenth of a single-pair
// OpticalFlow call.  Same reuse rule as the host pool.  The pool is bounded (SCANNER_SHIM_DEV_POOL_MB; default a quarter
// of the device's memory, at least 4 GB; 0 turns it off) and drained when the driver runs out of memory (a job that changes
// its batch or frame size, or shares the GPU with another allocator, must not die while idle blocks sit here).
// Ownership rule (Scanner's own for its pooled element buffers): whoever frees a buffer has finished with it -- a kernel's
// execute() returns with its inputs consumed and its outputs complete (every kernel class of this library synchronises
// its stream before it returns), so no queued work touches a block when it comes back here.  hipFree would ALSO have
// synchronised the whole device; a pooled block does not, deliberately: a device-wide wait at every free would serialise
// the concurrent kernel instances of one process (pipeline_instances_per_node > 1).  SCANNER_SHIM_DEV_POOL_SYNC=1
// restores hipFree's guarantee (hipDeviceSynchronize before a block is pooled) for code that does not keep the rule.
struct DevBlock { u8* p; size_t size; int device; };
std::vector<DevBlock> g_dev_pool;                   // free blocks
std::map<uintptr_t, DevBlock> g_dev_live;           // blocks handed out
size_t g_dev_pooled_bytes = 0;

size_t dev_pool_cap() {
  static const size_t cap = [] {
    const char* e = getenv("SCANNER_SHIM_DEV_POOL_MB");
    if (e && *e) {
      const long long mb = atoll(e);
      return (size_t)(mb < 0 ? 0 : mb) << 20;
    }
    // default: a quarter of the device's memory, at least 4 GB (the output of one 256-pair 1080p OpticalFlow call is
    // 4.25 GB: under a fixed 4 GB cap every such execute() went back to hipMalloc / hipFree and their device-wide sync)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = 0; }
    const size_t quarter = total_b / 4, floor_b = (size_t)4096 << 20;
    return quarter > floor_b ? quarter : floor_b;
  }();
  return cap;
}

// Returns every pooled block of `device` (any device if < 0) to the driver; the number of bytes released.
size_t drain_dev_pool(int device) {
  std::vector<DevBlock> out;
  {
    std::lock_guard<std::mutex> lk(g_mem_mutex);
    for (size_t i = 0; i < g_dev_pool.size();) {
      if (device < 0 || g_dev_pool[i].device == device) {
        out.push_back(g_dev_pool[i]);
        g_dev_pooled_bytes -= g_dev_pool[i].size;
        g_dev_pool.erase(g_dev_pool.begin() + i);
      } else {
        ++i;
      }
    }
  }
  size_t bytes = 0;
  for (auto& b : out) {
    (void)hipSetDevice(b.device);
    (void)hipFree(b.p);
    bytes += b.size;
  }
  return bytes;
}

// No drain from a static destructor: at that point the HIP runtime (and, in a Python process, torch's allocator) may already
// be tearing down, daemon threads may still be in raw_free, and a late hipFree is a known source of exit-time hangs.  The
// driver reclaims the pooled blocks when the process ends; a process that wants them back earlier calls
// stshim_dev_pool_drain (scannertools_amd.engine registers it with atexit, which runs before the runtime goes away).

u8* raw_alloc(DeviceHandle device, size_t size) {
  void* p = nullptr;
  if (size == 0) size = 1;
  if (device.type == DeviceType::GPU) {
    {
      std::lock_guard<std::mutex> lk(g_mem_mutex);
      for (size_t i = 0; i < g_dev_pool.size(); ++i) {
        if (g_dev_pool[i].device == device.id && g_dev_pool[i].size >= size && g_dev_pool[i].size <= 2 * size) {
          DevBlock b = g_dev_pool[i];
          g_dev_pool.erase(g_dev_pool.begin() + i);
          g_dev_pooled_bytes -= b.size;
          g_dev_live[(uintptr_t)b.p] = b;
          return b.p;
        }
      }
    }
    LOG_IF(FATAL, hipSetDevice(device.id) != hipSuccess) << "hipSetDevice failed";
    hipError_t e = hipMalloc(&p, size);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      // out of memory with idle blocks in the pool: give them back (this device's first, then every device's -- another
      // device's blocks do not help this allocation, but a second failure is fatal and must not be avoidable)
      if (drain_dev_pool(device.id) > 0) e = hipMalloc(&p, size);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)drain_dev_pool(-1);
        (void)hipSetDevice(device.id);
        e = hipMalloc(&p, size);
      }
    }
    LOG_IF(FATAL, e != hipSuccess) << "hipMalloc of " << size << " bytes failed: " << hipGetErrorString(e);
    std::lock_guard<std::mutex> lk(g_mem_mutex);
    g_dev_live[(uintptr_t)p] = DevBlock{(u8*)p, size, device.id};
  } else {
    p = host_alloc(size);
  }
  return (u8*)p;
}

void raw_free(DeviceHandle device, u8* p) {
  if (device.type == DeviceType::GPU) {
    DevBlock b{p, 0, device.id};
    bool pool = false;
    {
      std::lock_guard<std::mutex> lk(g_mem_mutex);
      auto it = g_dev_live.find((uintptr_t)p);
      if (it != g_dev_live.end()) {
        b = it->second;
        g_dev_live.erase(it);
        pool = b.size <= dev_pool_cap() && g_dev_pooled_bytes + b.size <= dev_pool_cap();
      }
    }
    (void)hipSetDevice(device.id);
    if (pool) {
      static const bool sync_on_free = [] { const char* e = getenv("SCANNER_SHIM_DEV_POOL_SYNC"); return e && *e && *e != '0'; }();
      if (sync_on_free) (void)hipDeviceSynchronize();  // hipFree's guarantee, on request (see above)
      std::lock_guard<std::mutex> lk(g_mem_mutex);
      g_dev_pool.push_back(b);
      g_dev_pooled_bytes += b.size;
      return;
    }
    (void)hipFree(p);
  } else {
    host_free(p);
  }
}
}  // namespace

size_t shim_dev_pool_bytes() {
  std::lock_guard<std::mutex> lk(g_mem_mutex);
  return g_dev_pooled_bytes;
}
size_t shim_dev_pool_drain(int device) { return drain_dev_pool(device); }

u8* new_block_buffer(DeviceHandle device, size_t size, i32 refs) {
  u8* p = raw_alloc(device, size);
  std::lock_guard<std::mutex> lk(g_mem_mutex);
  g_blocks[(uintptr_t)p] = Block{device, size, refs};
  return p;
}

u8* new_buffer(DeviceHandle device, size_t size) { return new_block_buffer(device, size, 1); }

void add_buffer_refs(DeviceHandle /*device*/, u8* buffer, size_t refs) {
  std::lock_guard<std::mutex> lk(g_mem_mutex);
  auto it = find_block(buffer);
  LOG_IF(FATAL, it == g_blocks.end()) << "add_buffer_ref on unknown buffer";
  it->second.refs += (i64)refs;
}

void add_buffer_ref(DeviceHandle device, u8* buffer) { add_buffer_refs(device, buffer, 1); }

void delete_buffer(DeviceHandle /*device*/, u8* buffer) {
  Block blk{};
  u8* base = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_mem_mutex);
    auto it = find_block(buffer);
    LOG_IF(FATAL, it == g_blocks.end()) << "delete_buffer on unknown buffer";
    if (--it->second.refs > 0) return;
    blk = it->second;
    base = (u8*)it->first;
    g_blocks.erase(it);
  }
  raw_free(blk.device, base);
}

void memcpy_buffer(u8* dest, DeviceHandle dest_device, const u8* src, DeviceHandle src_device, size_t size) {
  if (dest_device.type == DeviceType::CPU && src_device.type == DeviceType::CPU) {
    memcpy(dest, src, size);
    return;
  }
  hipMemcpyKind kind = dest_device.type == DeviceType::CPU   ? hipMemcpyDeviceToHost
                       : src_device.type == DeviceType::CPU ? hipMemcpyHostToDevice
                                                            : hipMemcpyDeviceToDevice;
  LOG_IF(FATAL, hipMemcpy(dest, src, size, kind) != hipSuccess) << "hipMemcpy failed";
}

size_t shim_live_buffers(DeviceType type) {
  std::lock_guard<std::mutex> lk(g_mem_mutex);
  size_t n = 0;
  for (auto& kv : g_blocks) n += kv.second.device.type == type;
  return n;
}

Frame* new_frame(DeviceHandle device, FrameInfo info) { return new Frame(info, new_buffer(device, info.size())); }

std::vector<Frame*> new_frames(DeviceHandle device, FrameInfo info, i32 num) {
  std::vector<Frame*> out;
  if (num <= 0) return out;
  u8* block = new_block_buffer(device, info.size() * (size_t)num, num);
  for (i32 i = 0; i < num; ++i) out.push_back(new Frame(info, block + info.size() * (size_t)i));
  return out;
}

void delete_frame(DeviceHandle device, Frame* frame) {
  delete_buffer(device, frame->data);
  delete frame;
}

// ---- registries ------------------------------------------------------------------------------
namespace {
std::vector<KernelRegistration>& kernel_registry() { static std::vector<KernelRegistration> r; return r; }
std::vector<OpRegistration>& op_registry() { static std::vector<OpRegistration> r; return r; }
}  // namespace

KernelBuilder::KernelBuilder(const std::string& op, KernelConstructor ctor, KernelKind kind) {
  reg_.op_name = op;
  reg_.constructor = std::move(ctor);
  reg_.kind = kind;
}
KernelBuilder::~KernelBuilder() {
  if (reg_.constructor) kernel_registry().push_back(reg_);
}
OpBuilder::~OpBuilder() {
  if (!reg_.name.empty()) op_registry().push_back(reg_);
}

}  // namespace scanner

// ---- mini engine + C API ---------------------------------------------------------------------
using namespace scanner;

namespace {

struct EngineKernel {
  KernelRegistration reg;
  const OpRegistration* op = nullptr;
  DeviceHandle device;
  Profiler profiler;  // what a Scanner worker hands every kernel instance (set_profiler)
  std::unique_ptr<BaseKernel> kernel;
};

struct EngineOutputs {
  DeviceHandle device;
  Elements elements;            // first output column: one element per input row
  std::vector<Elements> extra;  // further output columns of ops that declare several (CPM2)
};

Element* output_at(EngineOutputs* o, int col, int i) {
  if (!o || col < 0 || i < 0) return nullptr;
  Elements* e = col == 0 ? &o->elements : (col - 1 < (int)o->extra.size() ? &o->extra[col - 1] : nullptr);
  return e && i < (int)e->size() ? &(*e)[i] : nullptr;
}

// per calling thread: the engine entry points run concurrently from several threads (one per kernel instance), and each
// thread reads the figures of ITS last call
thread_local double g_last_execute_seconds = 0.0;  // wall time spent inside execute() by the last stshim_run_frames
// the same without the run's first execute() call (a fresh kernel instance allocates its device scratch there), and
// the rows those later calls covered: the steady state of a kernel instance that lives for a whole job
thread_local double g_last_steady_seconds = 0.0;
thread_local int g_last_steady_rows = 0;

void set_err(char* err, size_t n, const std::string& s) {
  if (err && n) { strncpy(err, s.c_str(), n - 1); err[n - 1] = 0; }
}

const OpRegistration* find_op(const char* name) {
  // the last registration of a name wins, like re-registering a module
  const OpRegistration* f = nullptr;
  for (auto& o : op_registry()) if (o.name == name) f = &o;
  return f;
}

}  // namespace

#define SHIM_EXPORT extern "C" __attribute__((visibility("default")))

SHIM_EXPORT int stshim_num_kernels() { return (int)kernel_registry().size(); }

SHIM_EXPORT int stshim_kernel_info(int i, char* op, int op_len, int* device_type, int* kind, int* can_batch) {
  if (i < 0 || i >= (int)kernel_registry().size()) return 1;
  const auto& r = kernel_registry()[i];
  if (op && op_len > 0) { strncpy(op, r.op_name.c_str(), op_len - 1); op[op_len - 1] = 0; }
  if (device_type) *device_type = (int)r.device_type;
  if (kind) *kind = (int)r.kind;
  if (can_batch) *can_batch = r.can_batch;
  return 0;
}

// n_in/n_out: column counts; stencil: up to max_stencil offsets written, count returned in n_stencil
SHIM_EXPORT int stshim_op_info(const char* op, int* n_in, int* n_out, int* out_is_frame, int* stencil, int max_stencil,
                               int* n_stencil) {
  const OpRegistration* o = find_op(op);
  if (!o) return 1;
  if (n_in) *n_in = (int)o->inputs.size();
  if (n_out) *n_out = (int)o->outputs.size();
  if (out_is_frame) *out_is_frame = !o->outputs.empty() && o->outputs[0].type == ColumnType::Video;
  int ns = (int)o->stencil.size();
  if (n_stencil) *n_stencil = ns;
  for (int i = 0; i < ns && i < max_stencil; ++i) stencil[i] = o->stencil[i];
  return 0;
}

SHIM_EXPORT void* stshim_kernel_create(const char* op, int device_type, int device_id, const uint8_t* args,
                                       size_t n_args, char* err, size_t err_len) {
  const KernelRegistration* found = nullptr;
  for (auto& r : kernel_registry())
    if (r.op_name == op && (int)r.device_type == device_type) found = &r;
  if (!found) {
    set_err(err, err_len, std::string("no kernel registered for op ") + op + " on that device type");
    return nullptr;
  }
  auto* ek = new EngineKernel();
  ek->reg = *found;
  ek->op = find_op(op);
  ek->device = DeviceHandle{(DeviceType)device_type, device_id};
  KernelConfig cfg;
  cfg.devices.push_back(ek->device);
  if (ek->op) {
    for (auto& c : ek->op->inputs) cfg.input_columns.push_back(c.name);
    for (auto& c : ek->op->outputs) cfg.output_columns.push_back(c.name);
  }
  // ops declared with stream_protobuf_name() receive their arguments per stream through
  // new_stream(), the others in KernelConfig::args (protobuf_name())
  const bool stream_args = ek->op && !ek->op->stream_protobuf_name.empty();
  std::vector<u8> arg_bytes;
  if (args && n_args) arg_bytes.assign(args, args + n_args);
  if (!stream_args) cfg.args = arg_bytes;
  ek->kernel.reset(found->constructor(cfg));
  ek->kernel->set_profiler(&ek->profiler);
  Result res;
  ek->kernel->validate(&res);
  if (!res.success()) {
    set_err(err, err_len, res.msg());
    delete ek;
    return nullptr;
  }
  ek->kernel->reset();
  if (stream_args) ek->kernel->new_stream(arg_bytes);
  return ek;
}

SHIM_EXPORT void stshim_kernel_destroy(void* k) { delete (EngineKernel*)k; }

// Runs the op over a stream of n frames.  frames[i]: pointer (on the kernel's device) to a dense
// (h,w,c) frame of `frame_type`.  batch: rows per execute() for batched kernels.  stencil /
// n_stencil: override of the op's registered stencil (n_stencil == 0 keeps it).  Rows whose
// stencil reaches outside [0,n) use the clamped edge frame.
SHIM_EXPORT void* stshim_run_frames(void* k, const void* const* frames, int n, int h, int w, int c, int frame_type,
                                    int batch, const int* stencil, int n_stencil, char* err, size_t err_len) {
  auto* ek = (EngineKernel*)k;
  if (!ek || n < 0 || (n > 0 && !frames)) { set_err(err, err_len, "bad arguments"); return nullptr; }
  std::vector<i32> st;
  if (n_stencil > 0) st.assign(stencil, stencil + n_stencil);
  else if (ek->op && !ek->op->stencil.empty()) st = ek->op->stencil;
  else st = {0};
  const bool stenciled = ek->reg.kind == KernelKind::Stenciled || ek->reg.kind == KernelKind::StenciledBatched;
  if (!stenciled && (st.size() != 1 || st[0] != 0)) {
    set_err(err, err_len, "stencil given to a kernel that is not stenciled");
    return nullptr;
  }
  const bool batched = ek->reg.kind == KernelKind::Batched || ek->reg.kind == KernelKind::StenciledBatched;
  if (batch < 1 || !batched) batch = 1;

  FrameInfo info(h, w, c, (FrameType)frame_type);
  std::vector<Frame> in_frames;
  in_frames.reserve(n);
  for (int i = 0; i < n; ++i) in_frames.emplace_back(info, (u8*)frames[i]);
  auto elem = [&](int row) {
    int rr = std::min(std::max(row, 0), n - 1);
    Element e(&in_frames[rr]);
    e.index = rr;
    return e;
  };

  auto* outs = new EngineOutputs();
  outs->device = ek->device;
  g_last_execute_seconds = 0.0;
  g_last_steady_seconds = 0.0;
  g_last_steady_rows = 0;
  for (int r0 = 0; r0 < n; r0 += batch) {
    const auto t_begin = std::chrono::steady_clock::now();
    const int nb = std::min(batch, n - r0);
    const size_t n_out = ek->op && ek->op->outputs.size() > 1 ? ek->op->outputs.size() : 1;
    BatchedElements out_cols(n_out);
    outs->extra.resize(n_out - 1);
    switch (ek->reg.kind) {
      case KernelKind::Batched: {
        BatchedElements in(1);
        for (int i = 0; i < nb; ++i) in[0].push_back(elem(r0 + i));
        static_cast<BatchedKernel*>(ek->kernel.get())->execute(in, out_cols);
        break;
      }
      case KernelKind::StenciledBatched: {
        StenciledBatchedElements in(1);
        for (int i = 0; i < nb; ++i) {
          Elements win;
          for (i32 s : st) win.push_back(elem(r0 + i + s));
          in[0].push_back(win);
        }
        static_cast<StenciledBatchedKernel*>(ek->kernel.get())->execute(in, out_cols);
        break;
      }
      case KernelKind::Stenciled: {
        StenciledElements in(1);
        for (i32 s : st) in[0].push_back(elem(r0 + s));
        Elements o;
        static_cast<StenciledKernel*>(ek->kernel.get())->execute(in, o);
        out_cols[0] = o;
        break;
      }
      case KernelKind::Plain: {
        Elements in{elem(r0)}, o;
        static_cast<Kernel*>(ek->kernel.get())->execute(in, o);
        out_cols[0] = o;
        break;
      }
    }
    {
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
      g_last_execute_seconds += dt;
      if (r0 > 0) { g_last_steady_seconds += dt; g_last_steady_rows += nb; }
    }
    for (auto& e : out_cols[0]) outs->elements.push_back(e);
    for (size_t c = 1; c < n_out; ++c)
      for (auto& e : out_cols[c]) outs->extra[c - 1].push_back(e);
    for (size_t c = 0; c < n_out; ++c)
      if ((int)out_cols[c].size() != nb) {
        set_err(err, err_len, "kernel produced " + std::to_string(out_cols[c].size()) + " outputs in column " + std::to_string(c) + " for " +
                                  std::to_string(nb) + " rows");
        return outs;  // the caller frees what was produced
      }
  }
  return outs;
}

// Several input columns (Batched / Plain kernels, no stencil): column c, row r is ptrs[c * n + r]; frame
// columns carry a dense frame of shape shapes[3c..3c+2] and type types[c], byte columns sizes[c * n + r]
// bytes.  Frame columns live on the kernel's device, byte columns where the caller put them (the
// reference's only multi-input op on this path reads its byte column on the host).  Returns the first
// output column.
SHIM_EXPORT void* stshim_run_columns(void* k, int n_cols, int n, const void* const* ptrs, const size_t* sizes,
                                     const int* is_frame, const int* shapes, const int* types, int batch, char* err,
                                     size_t err_len) {
  auto* ek = (EngineKernel*)k;
  if (!ek || n_cols < 1 || n < 0 || (n > 0 && !ptrs)) { set_err(err, err_len, "bad arguments"); return nullptr; }
  if (ek->reg.kind != KernelKind::Batched && ek->reg.kind != KernelKind::Plain) {
    set_err(err, err_len, "stshim_run_columns drives Batched and Plain kernels only");
    return nullptr;
  }
  if (batch < 1 || ek->reg.kind != KernelKind::Batched) batch = 1;
  std::vector<std::vector<Frame>> frames(n_cols);
  for (int c = 0; c < n_cols; ++c)
    if (is_frame[c]) {
      FrameInfo info(shapes[3 * c], shapes[3 * c + 1], shapes[3 * c + 2], (FrameType)types[c]);
      frames[c].reserve(n);
      for (int r = 0; r < n; ++r) frames[c].emplace_back(info, (u8*)ptrs[(size_t)c * n + r]);
    }
  auto elem = [&](int c, int r) {
    Element e = is_frame[c] ? Element(&frames[c][r]) : Element((u8*)ptrs[(size_t)c * n + r], sizes[(size_t)c * n + r]);
    e.index = r;
    return e;
  };
  auto* outs = new EngineOutputs();
  outs->device = ek->device;
  g_last_execute_seconds = 0.0;
  g_last_steady_seconds = 0.0;
  g_last_steady_rows = 0;
  for (int r0 = 0; r0 < n; r0 += batch) {
    const auto t_begin = std::chrono::steady_clock::now();
    const int nb = std::min(batch, n - r0);
    BatchedElements out_cols(1);
    if (ek->reg.kind == KernelKind::Batched) {
      BatchedElements in(n_cols);
      for (int c = 0; c < n_cols; ++c)
        for (int i = 0; i < nb; ++i) in[c].push_back(elem(c, r0 + i));
      static_cast<BatchedKernel*>(ek->kernel.get())->execute(in, out_cols);
    } else {
      Elements in, o;
      for (int c = 0; c < n_cols; ++c) in.push_back(elem(c, r0));
      static_cast<Kernel*>(ek->kernel.get())->execute(in, o);
      out_cols[0] = o;
    }
    {
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
      g_last_execute_seconds += dt;
      if (r0 > 0) { g_last_steady_seconds += dt; g_last_steady_rows += nb; }
    }
    for (auto& e : out_cols[0]) outs->elements.push_back(e);
    if ((int)out_cols[0].size() != nb) {
      set_err(err, err_len, "kernel produced " + std::to_string(out_cols[0].size()) + " outputs for " + std::to_string(nb) + " rows");
      return outs;
    }
  }
  return outs;
}

SHIM_EXPORT int stshim_outputs_count(void* o) { return o ? (int)((EngineOutputs*)o)->elements.size() : 0; }

SHIM_EXPORT int stshim_outputs_columns(void* o) { return o ? 1 + (int)((EngineOutputs*)o)->extra.size() : 0; }

SHIM_EXPORT int stshim_output_get_col(void* o, int col, int i, const void** data, size_t* size, int* is_frame, int* shape3, int* type) {
  Element* ep = output_at((EngineOutputs*)o, col, i);
  if (!ep) return 1;
  Element& e = *ep;
  if (e.is_frame) {
    Frame* f = e.as_frame();
    if (data) *data = f->data;
    if (size) *size = f->size();
    if (shape3) { shape3[0] = f->shape[0]; shape3[1] = f->shape[1]; shape3[2] = f->shape[2]; }
    if (type) *type = (int)f->type;
  } else {
    if (data) *data = e.buffer;
    if (size) *size = e.size;
  }
  if (is_frame) *is_frame = e.is_frame;
  return 0;
}

SHIM_EXPORT int stshim_output_get(void* o, int i, const void** data, size_t* size, int* is_frame, int* shape3, int* type) {
  return stshim_output_get_col(o, 0, i, data, size, is_frame, shape3, type);
}

SHIM_EXPORT int stshim_output_copy_col(void* o, int col, int i, void* dst_host, size_t n) {
  auto* outs = (EngineOutputs*)o;
  const void* p = nullptr;
  size_t sz = 0;
  if (stshim_output_get_col(o, col, i, &p, &sz, nullptr, nullptr, nullptr) || n > sz) return 1;
  memcpy_buffer((u8*)dst_host, CPU_DEVICE, (const u8*)p, outs->device, n);
  return 0;
}

SHIM_EXPORT int stshim_output_copy(void* o, int i, void* dst_host, size_t n) { return stshim_output_copy_col(o, 0, i, dst_host, n); }

SHIM_EXPORT void stshim_outputs_free(void* o) {
  auto* outs = (EngineOutputs*)o;
  if (!outs) return;
  auto drop = [&](Elements& col) {
    for (auto& e : col) {
      if (e.is_frame) delete_frame(outs->device, e.as_frame());
      else if (e.buffer) delete_buffer(outs->device, e.buffer);
    }
  };
  drop(outs->elements);
  for (auto& col : outs->extra) drop(col);
  delete outs;
}

SHIM_EXPORT double stshim_last_execute_seconds() { return g_last_execute_seconds; }

// Intervals the kernel instance has recorded under `key` through its Profiler: count, total seconds in *seconds.
SHIM_EXPORT int stshim_profiler_intervals(void* kernel, const char* key, double* seconds) {
  auto* ek = static_cast<EngineKernel*>(kernel);
  int n = 0;
  double total = 0;
  if (ek && key)
    for (auto& r : ek->profiler.get_records())
      if (r.key == key) { ++n; total += (double)(r.end - r.start) * 1e-9; }
  if (seconds) *seconds = total;
  return n;
}
SHIM_EXPORT double stshim_last_steady_seconds(int* rows) {
  if (rows) *rows = g_last_steady_rows;
  return g_last_steady_seconds;
}

SHIM_EXPORT size_t stshim_live_buffers(int device_type) { return shim_live_buffers((DeviceType)device_type); }

// Device-buffer pool: bytes idle in it / give them back to the driver (all devices if device < 0); returns bytes released.
SHIM_EXPORT size_t stshim_dev_pool_bytes() { return shim_dev_pool_bytes(); }
SHIM_EXPORT size_t stshim_dev_pool_drain(int device) { return shim_dev_pool_drain(device); }
