// Internal declarations shared by the translation units of libscannertools_hip.so.
#ifndef ST_INTERNAL_H_
#define ST_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "scannertools_hip.h"

#define ST_EXPORT extern "C" __attribute__((visibility("default")))

struct st_timing_slot {
  std::vector<hipEvent_t> starts, stops;  // pooled events, one pair per recorded launch
  size_t used = 0;
  int launches = 0;
  double total_ms = 0.0;
};

struct st_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // flow-iteration kernel choice for small launches (ST_ITER_TILE, read when the context is created):
  // -1 by total size (default), 0 never the tile kernel, 1 always.  The two kernels agree bit for
  // bit, so this is a scheduling switch only.
  int tile_mode = -1;
  long long tile_px = 600000;
  bool poly_u8 = true;     // level-0 expansion straight from the gray frames (k_polyexp_u8), level 0 left out of the pyramid pass; ST_POLY_U8=0: float source
  bool fold_gray = false;  // ST_PYR_FOLD_GRAY=1: luma conversion inside the one-pass pyramid (slower; A/B switch)
  // role-split kernels (scheduling switches like tile_mode, read when the context is created; results do not depend on them):
  // ST_ITER_ROLES / ST_PYR_ROLES: -1 by launch size (default), 0 never, 1 always; ST_ROLES_NCW: 0 = by cost, 4 or 5 column waves
  int roles_mode = -1, roles_ncw = 0, roles_rows = 0, pyr_roles = -1;
  int conv_tile = -1;   // ST_CONV_TILE: 0 = bf16x3 convolutions always on the per-tap kernel (st_conv.hip)
  // Concurrent kernel instances (Scanner's pipeline_instances_per_node: K contexts of one process on one GPU, each call followed
  // by st_ctx_sync).  flow_busy: an OpticalFlow call of this context has been enqueued and not yet synchronised; flow_enter_ns:
  // when (steady clock).  st_flow_call_begins() reads the other contexts' to pick kernels that share the chip (st_context.hip).
  std::atomic<bool> flow_busy{false};
  std::atomic<long long> flow_enter_ns{0};
  bool flow_concurrent = false;   // decided at the entry of the current st_farneback_pairs call
  int concurrency_mode = -1;      // ST_CONCURRENT: -1 detect (default), 0 never assume, 1 always assume other instances
  // bump-allocated scratch
  void* ws = nullptr;
  size_t ws_bytes = 0;
  size_t ws_limit = (size_t)64 << 30;
  size_t ws_off = 0;
  // small device table for pointer arrays
  unsigned timing_mask = 0;
  st_timing_slot timing[ST_K_COUNT];
  std::string last_error;
  int num_cus = 256;
};

int st_set_error(st_ctx* ctx, int status, const char* fmt, ...);

#define ST_HIP(ctx, expr)                                                                   \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      return st_set_error((ctx), ST_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                \
                          hipGetErrorString(_e), __FILE__, __LINE__);                       \
  } while (0)

#define ST_TRY(expr)              \
  do {                            \
    int _s = (expr);              \
    if (_s != ST_OK) return _s;   \
  } while (0)

// Enter an API call: null check + select device.
int st_enter(st_ctx* ctx);

// Marks `ctx` as having an OpticalFlow call in flight and says whether TWO OR MORE other contexts of this process have one in
// flight on the same device (entered within the last 50 ms and not synchronised since): then a small launch should not take a
// whole CU per workgroup.  A scheduling decision only -- the kernels it chooses between agree bit for bit.
bool st_flow_call_begins(st_ctx* ctx);

// Scratch: reset at the start of a call, then bump-allocate (256-B aligned).  Grows the
// backing allocation when needed (synchronising the stream first).
int st_ws_reserve(st_ctx* ctx, size_t total_bytes);
void st_ws_reset(st_ctx* ctx);
void* st_ws_alloc(st_ctx* ctx, size_t bytes);  // nullptr if the reservation is exhausted
inline size_t st_align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Timing brackets for kernel class `id` (no-ops unless enabled in timing_mask).
int st_time_begin(st_ctx* ctx, int id);
int st_time_end(st_ctx* ctx, int id);
int st_time_dispatch(st_ctx* ctx, int id, hipEvent_t* start, hipEvent_t* stop);   // events for hipExtLaunchKernelGGL (st_context.hip)

// A bracket is closed only if it was opened: when st_time_begin fails (event creation / record) the
// launch simply goes untimed; the failure stays in last_error and the slot's counters are untouched.
struct st_timed {
  st_ctx* ctx;
  int id;
  bool open;
  st_timed(st_ctx* c, int k) : ctx(c), id(k), open(st_time_begin(c, k) == ST_OK) {}
  ~st_timed() { if (open) (void)st_time_end(ctx, id); }
};

// A pointer read from a device-side pointer table (Scanner hands every frame as its own buffer) has no provable
// address space, so loads and stores through it compile to flat_* instructions: those also count on lgkmcnt (an LDS
// wait then waits for them too) and take a 64-bit address in VGPRs.  Everything this library is handed lives in
// global memory; the round trip through address space 1 tells the compiler so (global_* instructions, SGPR base).
#ifdef __HIPCC__
template <class T>
__device__ __forceinline__ T* st_gl(T* p) {
  return (T*)((__attribute__((address_space(1))) T*)(uintptr_t)p);
}
#endif

#endif  // ST_INTERNAL_H_
