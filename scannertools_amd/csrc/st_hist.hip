// Per-channel histogram of interleaved U8x3 frames.
//
// Replaces the arithmetic of HistogramKernelCPU::execute
// (/root/reference/scannertools/scannertools_cpp/imgproc/histogram_kernel_cpu.cpp:25-45):
// three cv::calcHist passes per frame become ONE pass over the frame's bytes.
//
// Each frame byte is read exactly once with 16-B-per-lane coalesced loads; counts go to LDS
// sub-histograms with ds_add_u32 (no return value), are folded to `bins` and committed with one
// global atomic per non-empty bin per workgroup.  Algorithmic bytes per frame = 3*w*h + 3*bins*4.
// Kernels: k_hist_u8c3_v2<32,1024> (default: one 1024-thread workgroup per CU, 32 lane-indexed
// copies of the 768 counters = 96 KB of LDS, so a lane always hits its own bank and no two lanes of
// an LDS pass ever meet on a counter: data-independent throughput), k_hist_u8c3_v2<8,256> and
// k_hist_u8c3 (one copy per wave) kept for A/B runs (ST_HIST_VARIANT=8 / 0).
// Measured at 256 1080p frames per launch: 5.7 TB/s (scripts/bench_hist.py); LDS-atomic rates per
// layout: scripts/ubench/ldsatomic.hip.
#include <cstdlib>
#include <string>

#include <hip/hip_ext.h>

#include "st_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

struct FrameSrc {
  const uint8_t* const* ptrs;  // device table of frame pointers, or null
  const uint8_t* base;         // strided stream
  size_t stride;
};

__device__ __forceinline__ void lds_inc(unsigned* p) {
  __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Count the 16 bytes of q; c0/c1/c2 are the sub-histogram offsets (channel*256) of bytes
// 0,1,2 (mod 3) of this vector.
__device__ __forceinline__ void count16(unsigned* h, uint4 q, unsigned c0, unsigned c1, unsigned c2) {
  lds_inc(h + c0 + (q.x & 0xff));
  lds_inc(h + c1 + ((q.x >> 8) & 0xff));
  lds_inc(h + c2 + ((q.x >> 16) & 0xff));
  lds_inc(h + c0 + (q.x >> 24));
  lds_inc(h + c1 + (q.y & 0xff));
  lds_inc(h + c2 + ((q.y >> 8) & 0xff));
  lds_inc(h + c0 + ((q.y >> 16) & 0xff));
  lds_inc(h + c1 + (q.y >> 24));
  lds_inc(h + c2 + (q.z & 0xff));
  lds_inc(h + c0 + ((q.z >> 8) & 0xff));
  lds_inc(h + c1 + ((q.z >> 16) & 0xff));
  lds_inc(h + c2 + (q.z >> 24));
  lds_inc(h + c0 + (q.w & 0xff));
  lds_inc(h + c1 + ((q.w >> 8) & 0xff));
  lds_inc(h + c2 + ((q.w >> 16) & 0xff));
  lds_inc(h + c0 + (q.w >> 24));
}

__global__ __launch_bounds__(kThreads) void k_hist_u8c3(FrameSrc src, long long nbytes, int chunks, int bins,
                                                        int32_t* __restrict__ out) {
  __shared__ unsigned sh[kWaves * 768];
  const int tid = threadIdx.x;
  const int frame = blockIdx.y;
  const int chunk = blockIdx.x;
  const uint8_t* p = src.ptrs ? src.ptrs[frame] : src.base + (size_t)frame * src.stride;

  for (int i = tid; i < kWaves * 768; i += kThreads) sh[i] = 0;
  __syncthreads();
  unsigned* my = sh + (tid >> 6) * 768;

  // [0, head) and [tail, nbytes) are the unaligned ends; [head, tail) is read as uint4.
  long long head = (long long)((16 - ((uintptr_t)p & 15)) & 15);
  if (head > nbytes) head = nbytes;
  const long long nvec = (nbytes - head) >> 4;
  const long long tail = head + (nvec << 4);
  const uint4* vp = reinterpret_cast<const uint4*>(p + head);

  const long long per = (nvec + chunks - 1) / chunks;
  const long long v0 = (long long)chunk * per;
  long long v1 = v0 + per;
  if (v1 > nvec) v1 = nvec;

  // Channel of byte j of vector i is (head + 16 i + j) % 3 = (head + i + j) % 3.  A thread's
  // vector index advances by 256 = 1 (mod 3) per step, so three steps cycle the phase.
  long long i = v0 + tid;
  unsigned ph = (unsigned)((head + i) % 3);
  const unsigned o0 = ph * 256, o1 = ((ph + 1) % 3) * 256, o2 = ((ph + 2) % 3) * 256;
  for (; i + 2 * kThreads < v1; i += 3 * kThreads) {
    uint4 a = vp[i], b = vp[i + kThreads], c = vp[i + 2 * kThreads];
    count16(my, a, o0, o1, o2);
    count16(my, b, o1, o2, o0);
    count16(my, c, o2, o0, o1);
  }
  if (i < v1) {
    uint4 a = vp[i];
    count16(my, a, o0, o1, o2);
    if (i + kThreads < v1) {
      uint4 b = vp[i + kThreads];
      count16(my, b, o1, o2, o0);
    }
  }
  if (chunk == 0) {
    for (long long b = tid; b < head; b += kThreads) lds_inc(my + (unsigned)(b % 3) * 256 + p[b]);
    for (long long b = tail + tid; b < nbytes; b += kThreads) lds_inc(my + (unsigned)(b % 3) * 256 + p[b]);
  }
  __syncthreads();

  int32_t* o = out + (size_t)frame * 3 * bins;
  if (bins == 256) {
    for (int b = tid; b < 768; b += kThreads) {
      unsigned s = sh[b] + sh[768 + b] + sh[2 * 768 + b] + sh[3 * 768 + b];
      if (s) atomicAdd(reinterpret_cast<unsigned*>(o) + b, s);
    }
  } else {
    // fold the four copies into copy 0, then one thread per output bin sums its value range
    for (int b = tid; b < 768; b += kThreads) sh[b] += sh[768 + b] + sh[2 * 768 + b] + sh[3 * 768 + b];
    __syncthreads();
    for (int ob = tid; ob < 3 * bins; ob += kThreads) {
      const int ch = ob / bins, bin = ob - ch * bins;
      // values v with floor(v*bins/256) == bin: v in [ceil(256*bin/bins), ceil(256*(bin+1)/bins))
      const int lo = (256 * bin + bins - 1) / bins, hi = (256 * (bin + 1) + bins - 1) / bins;
      unsigned s = 0;
      for (int v = lo; v < hi; ++v) s += sh[ch * 256 + v];
      if (s) atomicAdd(reinterpret_cast<unsigned*>(o) + ob, s);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// v2: C lane-indexed copies of the 3x256 counters shared by the whole workgroup.  A lane always
// updates copy (lane & (C-1)), so within a 32-lane LDS pass at most 32/C lanes can meet on one
// address and the bank pattern is (bin * C + lane) % 32: random data costs ~1.5x (C = 16) instead
// of the ~3.5x of one histogram per wave, and the all-equal frame costs 32/C-way instead of 32-way.
// ---------------------------------------------------------------------------------------------
// SHIFT: counters are kept for byte >> SHIFT (0: all 256 values; 4: the reference's 16 bins directly, histogram_kernel_cpu.cpp:8)
template <int C, int SHIFT = 0>
__device__ __forceinline__ void count16c(unsigned* h, uint4 q, unsigned c0, unsigned c1, unsigned c2) {
  // c0/c1/c2 already include the lane's copy index; counters of a bin are C dwords apart
  constexpr unsigned MK = 0xffu >> SHIFT;
  lds_inc(h + c0 + ((q.x >> SHIFT) & MK) * C);
  lds_inc(h + c1 + ((q.x >> (8 + SHIFT)) & MK) * C);
  lds_inc(h + c2 + ((q.x >> (16 + SHIFT)) & MK) * C);
  lds_inc(h + c0 + (q.x >> (24 + SHIFT)) * C);
  lds_inc(h + c1 + ((q.y >> SHIFT) & MK) * C);
  lds_inc(h + c2 + ((q.y >> (8 + SHIFT)) & MK) * C);
  lds_inc(h + c0 + ((q.y >> (16 + SHIFT)) & MK) * C);
  lds_inc(h + c1 + (q.y >> (24 + SHIFT)) * C);
  lds_inc(h + c2 + ((q.z >> SHIFT) & MK) * C);
  lds_inc(h + c0 + ((q.z >> (8 + SHIFT)) & MK) * C);
  lds_inc(h + c1 + ((q.z >> (16 + SHIFT)) & MK) * C);
  lds_inc(h + c2 + (q.z >> (24 + SHIFT)) * C);
  lds_inc(h + c0 + ((q.w >> SHIFT) & MK) * C);
  lds_inc(h + c1 + ((q.w >> (8 + SHIFT)) & MK) * C);
  lds_inc(h + c2 + ((q.w >> (16 + SHIFT)) & MK) * C);
  lds_inc(h + c0 + (q.w >> (24 + SHIFT)) * C);
}

// How a workgroup's counters reach the output row of its frame:
//   HC_ATOMIC  one global atomic per non-empty bin into a row the launcher zeroed (a memset launch ahead of the kernel);
//   HC_DIRECT  the frame has ONE workgroup (chunks == 1, launches of >= one frame per CU): plain stores, no memset.
// (Round 4 also built a third way for the chunked case -- partial rows in the workspace, a ticket per frame, the workgroup that
// draws the last ticket sums the rows and stores: no memset, no atomics on the output.  Measured at 32 frames per launch:
// with device-scope fences around the ticket 131 us instead of 36 (each fence writes back and invalidates its XCD's whole L2);
// with fence-free device-scope atomic accesses to the rows 37.8 us of kernel instead of 35.6 and 38-41 us per call instead of
// 37 -- the store acknowledgements and the ticket's round trip at the end of every workgroup cost more than the memset launch
// they replace; the 16-bin instance with its 128 chunks per frame 90 us.  Removed.)
enum { HC_ATOMIC = 0, HC_DIRECT = 1 };
struct HistCommit {
  int mode;
};

template <int C, int T, int SHIFT = 0, bool HALF = false>
__global__ __launch_bounds__(T) void k_hist_u8c3_v2(FrameSrc src, long long nbytes, int chunks, int bins,
                                                           int32_t* __restrict__ out, HistCommit hc) {
  static_assert(T % 3 == 1, "the channel phase of a thread's vectors must advance by one per step");
  constexpr int NB = 256 >> SHIFT;   // counters per channel
  __shared__ unsigned sh[3 * NB * C];
  const int tid = threadIdx.x;
  const int frame = blockIdx.y;
  const int chunk = blockIdx.x;
  const uint8_t* p = src.ptrs ? src.ptrs[frame] : src.base + (size_t)frame * src.stride;
  const unsigned copy = tid & (C - 1);

  long long head = (long long)((16 - ((uintptr_t)p & 15)) & 15);
  if (head > nbytes) head = nbytes;
  const long long nvec = (nbytes - head) >> 4;
  const long long tail = head + (nvec << 4);
  // every byte is read once: non-temporal loads (no L1 allocation, early eviction further out) -- measured +7-8 % (64 frames
  // per launch: 5.07 -> 5.54 TB/s)
  typedef unsigned u4nt __attribute__((ext_vector_type(4)));
  const u4nt* vq = reinterpret_cast<const u4nt*>(p + head);
  auto ld = [&](long long j) { const u4nt v = __builtin_nontemporal_load(vq + j); return make_uint4(v.x, v.y, v.z, v.w); };
  // vectors per chunk: a whole number of pipelined steps (6 T vectors), so that only the frame's last chunk runs the
  // un-pipelined remainder loops below
  const long long per = ((nvec + chunks - 1) / chunks + 6 * T - 1) / (6 * T) * (6 * T);
  const long long v0 = (long long)chunk * per;
  long long v1 = v0 + per;
  if (v1 > nvec) v1 = nvec;

  long long i = v0 + tid;
  const unsigned ph = (unsigned)((head + i) % 3);
  const unsigned o0 = ph * NB * C + copy, o1 = ((ph + 1) % 3) * NB * C + copy, o2 = ((ph + 2) % 3) * NB * C + copy;
  // Software pipeline over the steps every thread of the workgroup takes in full (six vectors =
  // two phase cycles per step): the loads of step k+1 are requested before the 96 counter updates
  // of step k.  The trip count is uniform and the loop has no conditional loads, so the compiler's
  // vmcnt bookkeeping sees the same number of outstanding loads on every path into the loop.
  const int full = v1 > v0 ? (int)((v1 - v0) / (6 * T)) : 0;
  // the first step's loads are requested BEFORE the counters are cleared: the clear (96 KB of LDS stores) and its
  // barrier then run under the first memory round trip instead of ahead of it (matters for short launches)
  uint4 a = {}, b = {}, c = {}, d = {}, e = {}, f = {};
  if (full > 0) {
    a = ld(i); b = ld(i + T); c = ld(i + 2 * T);
    d = ld(i + 3 * T); e = ld(i + 4 * T); f = ld(i + 5 * T);
  }
  for (int z = tid; z < 3 * NB * C; z += T) sh[z] = 0;
  __syncthreads();
  if (full > 0 && HALF) {
    // half-steps: the three vectors just counted are re-requested at once, so three to six are in flight and the drain after the
    // last load is three vectors' counting instead of six
    for (int it = 1; it < full; ++it) {
      i += 6 * T;
      count16c<C, SHIFT>(sh, a, o0, o1, o2);
      count16c<C, SHIFT>(sh, b, o1, o2, o0);
      count16c<C, SHIFT>(sh, c, o2, o0, o1);
      __builtin_amdgcn_sched_barrier(0);
      a = ld(i); b = ld(i + T); c = ld(i + 2 * T);
      __builtin_amdgcn_sched_barrier(0);
      count16c<C, SHIFT>(sh, d, o0, o1, o2);
      count16c<C, SHIFT>(sh, e, o1, o2, o0);
      count16c<C, SHIFT>(sh, f, o2, o0, o1);
      __builtin_amdgcn_sched_barrier(0);
      d = ld(i + 3 * T); e = ld(i + 4 * T); f = ld(i + 5 * T);
      __builtin_amdgcn_sched_barrier(0);
    }
    count16c<C, SHIFT>(sh, a, o0, o1, o2);
    count16c<C, SHIFT>(sh, b, o1, o2, o0);
    count16c<C, SHIFT>(sh, c, o2, o0, o1);
    count16c<C, SHIFT>(sh, d, o0, o1, o2);
    count16c<C, SHIFT>(sh, e, o1, o2, o0);
    count16c<C, SHIFT>(sh, f, o2, o0, o1);
    i += 6 * T;
  } else if (full > 0) {
    for (int it = 1; it < full; ++it) {
      i += 6 * T;
      const uint4 na = ld(i), nb = ld(i + T), nc = ld(i + 2 * T);
      const uint4 nd = ld(i + 3 * T), ne = ld(i + 4 * T), nf = ld(i + 5 * T);
      __builtin_amdgcn_sched_barrier(0);
      count16c<C, SHIFT>(sh, a, o0, o1, o2);
      count16c<C, SHIFT>(sh, b, o1, o2, o0);
      count16c<C, SHIFT>(sh, c, o2, o0, o1);
      count16c<C, SHIFT>(sh, d, o0, o1, o2);
      count16c<C, SHIFT>(sh, e, o1, o2, o0);
      count16c<C, SHIFT>(sh, f, o2, o0, o1);
      a = na; b = nb; c = nc; d = nd; e = ne; f = nf;
    }
    count16c<C, SHIFT>(sh, a, o0, o1, o2);
    count16c<C, SHIFT>(sh, b, o1, o2, o0);
    count16c<C, SHIFT>(sh, c, o2, o0, o1);
    count16c<C, SHIFT>(sh, d, o0, o1, o2);
    count16c<C, SHIFT>(sh, e, o1, o2, o0);
    count16c<C, SHIFT>(sh, f, o2, o0, o1);
    i += 6 * T;
  }
  for (; i + 2 * T < v1; i += 3 * T) {
    uint4 a = ld(i), b = ld(i + T), c = ld(i + 2 * T);
    count16c<C, SHIFT>(sh, a, o0, o1, o2);
    count16c<C, SHIFT>(sh, b, o1, o2, o0);
    count16c<C, SHIFT>(sh, c, o2, o0, o1);
  }
  if (i < v1) {
    uint4 a = ld(i);
    count16c<C, SHIFT>(sh, a, o0, o1, o2);
    if (i + T < v1) {
      uint4 b = ld(i + T);
      count16c<C, SHIFT>(sh, b, o1, o2, o0);
    }
  }
  if (chunk == 0) {
    for (long long b = tid; b < head; b += T) lds_inc(sh + ((unsigned)(b % 3) * NB + (p[b] >> SHIFT)) * C + copy);
    for (long long b = tail + tid; b < nbytes; b += T) lds_inc(sh + ((unsigned)(b % 3) * NB + (p[b] >> SHIFT)) * C + copy);
  }
  __syncthreads();
  // fold the C copies of every counter into copy 0
  for (int b = tid; b < 3 * NB; b += T) {
    unsigned s = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) s += sh[b * C + ((c + tid) & (C - 1))];
    sh[b * C] = s;  // only this thread touches bin b's copies
  }
  __syncthreads();
  int32_t* o = out + (size_t)frame * 3 * bins;
  const bool plain = hc.mode != HC_ATOMIC;
  if (SHIFT) {
    // the counters ARE the output bins (bins == NB, checked by the launcher)
    for (int ob = tid; ob < 3 * NB; ob += T) {
      const unsigned v = sh[ob * C];
      if (plain) o[ob] = (int32_t)v;
      else if (v) atomicAdd(reinterpret_cast<unsigned*>(o) + ob, v);
    }
    return;
  }
  for (int ob = tid; ob < 3 * bins; ob += T) {
    const int ch = ob / bins, bin = ob - ch * bins;
    const int lo = (256 * bin + bins - 1) / bins, hi = (256 * (bin + 1) + bins - 1) / bins;
    unsigned s = 0;
    for (int v = lo; v < hi; ++v) s += sh[(ch * 256 + v) * C];
    if (plain) o[ob] = (int32_t)s;
    else if (s) atomicAdd(reinterpret_cast<unsigned*>(o) + ob, s);
  }
}

// ---------------------------------------------------------------------------------------------
// p2: the v2 kernel with the counters of channels 0 and 1 sharing a dword (low / high 16 bits; channel 2 keeps whole dwords): 32
// lane-indexed copies then take 64 KB instead of 96 KB and TWO 1024-thread workgroups fit a CU (64 registers per thread), so
// one workgroup's first round trip, clear, fold and commit can run under the other's streaming.  Opt-in (ST_HIST_P2=1): the
// gain at 32-64 frames per launch is 1-4 % (the two workgroups of a CU start and finish together in a one-round launch, so
// little overlaps), see hist_launch.  Which half a byte's count goes to depends only on the byte's position in the vector, i.e. on
// the thread's phase: the increment (1 or 1 << 16) is a per-thread constant and the inner loop has the same instructions as v2's.
// A half-counter must stay below 65 536: a copy gets at most 6 bytes of a channel per vector of its T / 32 threads, so a
// chunk is limited to kP2MaxVec vectors (the launcher raises `chunks` accordingly); the fold widens to 32 bits.
// ---------------------------------------------------------------------------------------------
constexpr long long kP2MaxVec = 336000;   // 6 * 336000 / 32 = 63 000 (+ the frame's < 32 unaligned end bytes) < 65 536

__device__ __forceinline__ void lds_add(unsigned* p, unsigned v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int C>
__device__ __forceinline__ void count16p(unsigned* h, uint4 q, unsigned c0, unsigned c1, unsigned c2, unsigned i0, unsigned i1,
                                         unsigned i2) {
  lds_add(h + c0 + (q.x & 0xff) * C, i0);
  lds_add(h + c1 + ((q.x >> 8) & 0xff) * C, i1);
  lds_add(h + c2 + ((q.x >> 16) & 0xff) * C, i2);
  lds_add(h + c0 + (q.x >> 24) * C, i0);
  lds_add(h + c1 + (q.y & 0xff) * C, i1);
  lds_add(h + c2 + ((q.y >> 8) & 0xff) * C, i2);
  lds_add(h + c0 + ((q.y >> 16) & 0xff) * C, i0);
  lds_add(h + c1 + (q.y >> 24) * C, i1);
  lds_add(h + c2 + (q.z & 0xff) * C, i2);
  lds_add(h + c0 + ((q.z >> 8) & 0xff) * C, i0);
  lds_add(h + c1 + ((q.z >> 16) & 0xff) * C, i1);
  lds_add(h + c2 + (q.z >> 24) * C, i2);
  lds_add(h + c0 + (q.w & 0xff) * C, i0);
  lds_add(h + c1 + ((q.w >> 8) & 0xff) * C, i1);
  lds_add(h + c2 + ((q.w >> 16) & 0xff) * C, i2);
  lds_add(h + c0 + (q.w >> 24) * C, i0);
}

template <int C, int T>
__global__ __launch_bounds__(T, 8) void k_hist_u8c3_p2(FrameSrc src, long long nbytes, int chunks, int bins,
                                                        int32_t* __restrict__ out) {
  static_assert(T % 3 == 1, "the channel phase of a thread's vectors must advance by one per step");
  __shared__ unsigned sh[2 * 256 * C];   // [0, 256 C): channel 0 (low half) | channel 1 (high half); [256 C, 512 C): channel 2
  const int tid = threadIdx.x;
  const int frame = blockIdx.y;
  const int chunk = blockIdx.x;
  const uint8_t* p = src.ptrs ? src.ptrs[frame] : src.base + (size_t)frame * src.stride;
  const unsigned copy = tid & (C - 1);

  long long head = (long long)((16 - ((uintptr_t)p & 15)) & 15);
  if (head > nbytes) head = nbytes;
  const long long nvec = (nbytes - head) >> 4;
  const long long tail = head + (nvec << 4);
  typedef unsigned u4nt __attribute__((ext_vector_type(4)));
  const u4nt* vq = reinterpret_cast<const u4nt*>(p + head);
  auto ld = [&](long long j) { const u4nt v = __builtin_nontemporal_load(vq + j); return make_uint4(v.x, v.y, v.z, v.w); };
  const long long per = ((nvec + chunks - 1) / chunks + 6 * T - 1) / (6 * T) * (6 * T);
  const long long v0 = (long long)chunk * per;
  long long v1 = v0 + per;
  if (v1 > nvec) v1 = nvec;

  long long i = v0 + tid;
  const unsigned ph = (unsigned)((head + i) % 3);
  // offset and increment of channel ch
  auto off = [&](unsigned ch) { return (ch == 2 ? 256u * C : 0u) + copy; };
  auto inc = [&](unsigned ch) { return ch == 1 ? 0x10000u : 1u; };
  const unsigned o0 = off(ph), o1 = off((ph + 1) % 3), o2 = off((ph + 2) % 3);
  const unsigned i0 = inc(ph), i1 = inc((ph + 1) % 3), i2 = inc((ph + 2) % 3);
  const int full = v1 > v0 ? (int)((v1 - v0) / (6 * T)) : 0;
  uint4 a = {}, b = {}, c = {}, d = {}, e = {}, f = {};
  if (full > 0) {
    a = ld(i); b = ld(i + T); c = ld(i + 2 * T);
    d = ld(i + 3 * T); e = ld(i + 4 * T); f = ld(i + 5 * T);
  }
  {
    uint4* z4 = reinterpret_cast<uint4*>(sh);
    for (int z = tid; z < 2 * 256 * C / 4; z += T) z4[z] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  if (full > 0) {
    for (int it = 1; it < full; ++it) {
      i += 6 * T;
      const uint4 na = ld(i), nb = ld(i + T), nc = ld(i + 2 * T);
      const uint4 nd = ld(i + 3 * T), ne = ld(i + 4 * T), nf = ld(i + 5 * T);
      __builtin_amdgcn_sched_barrier(0);
      count16p<C>(sh, a, o0, o1, o2, i0, i1, i2);
      count16p<C>(sh, b, o1, o2, o0, i1, i2, i0);
      count16p<C>(sh, c, o2, o0, o1, i2, i0, i1);
      count16p<C>(sh, d, o0, o1, o2, i0, i1, i2);
      count16p<C>(sh, e, o1, o2, o0, i1, i2, i0);
      count16p<C>(sh, f, o2, o0, o1, i2, i0, i1);
      a = na; b = nb; c = nc; d = nd; e = ne; f = nf;
    }
    count16p<C>(sh, a, o0, o1, o2, i0, i1, i2);
    count16p<C>(sh, b, o1, o2, o0, i1, i2, i0);
    count16p<C>(sh, c, o2, o0, o1, i2, i0, i1);
    count16p<C>(sh, d, o0, o1, o2, i0, i1, i2);
    count16p<C>(sh, e, o1, o2, o0, i1, i2, i0);
    count16p<C>(sh, f, o2, o0, o1, i2, i0, i1);
    i += 6 * T;
  }
  for (; i + 2 * T < v1; i += 3 * T) {
    uint4 a = ld(i), b = ld(i + T), c = ld(i + 2 * T);
    count16p<C>(sh, a, o0, o1, o2, i0, i1, i2);
    count16p<C>(sh, b, o1, o2, o0, i1, i2, i0);
    count16p<C>(sh, c, o2, o0, o1, i2, i0, i1);
  }
  if (i < v1) {
    uint4 a = ld(i);
    count16p<C>(sh, a, o0, o1, o2, i0, i1, i2);
    if (i + T < v1) {
      uint4 b = ld(i + T);
      count16p<C>(sh, b, o1, o2, o0, i1, i2, i0);
    }
  }
  if (chunk == 0) {
    for (long long b = tid; b < head; b += T) { const unsigned ch = (unsigned)(b % 3); lds_add(sh + off(ch) + p[b] * C, inc(ch)); }
    for (long long b = tail + tid; b < nbytes; b += T) { const unsigned ch = (unsigned)(b % 3); lds_add(sh + off(ch) + p[b] * C, inc(ch)); }
  }
  __syncthreads();
  // fold the C copies: thread b < 256 unpacks value b of channels 0 and 1 into dwords 0 and 1 of its row, thread 256 + b sums
  // channel 2's row into its dword 0 (a row is touched by its thread only)
  if (tid < 512) {
    unsigned* row = sh + tid * C;
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const unsigned v = row[(c + tid) & (C - 1)];
      if (tid < 256) { lo += v & 0xffffu; hi += v >> 16; } else lo += v;
    }
    row[0] = lo;
    if (tid < 256) row[1] = hi;
  }
  __syncthreads();
  int32_t* o = out + (size_t)frame * 3 * bins;
  for (int ob = tid; ob < 3 * bins; ob += T) {
    const int ch = ob / bins, bin = ob - ch * bins;
    const int lo = (256 * bin + bins - 1) / bins, hi = (256 * (bin + 1) + bins - 1) / bins;
    const unsigned* col = sh + (ch == 2 ? 256 * C : ch);   // value v of channel ch: col[v * C]
    unsigned s = 0;
    for (int v = lo; v < hi; ++v) s += col[v * C];
    if (s) atomicAdd(reinterpret_cast<unsigned*>(o) + ob, s);
  }
}

// Chunks per frame for the default kernel (one 1024-thread workgroup per CU at a time): the
// launch takes ceil(n*c / CUs) rounds of workgroups that each stream 1/c of a frame, so pick the
// c that minimises rounds/c (ties: fewer, larger chunks), with at least 6 vectors (one pipelined step) per thread.
long long chunks_for(int num_cus, int n, long long nvec, int threads) {
  long long mx = (nvec + 3LL * threads * 2 - 1) / (3LL * threads * 2);
  if (mx < 1) mx = 1;
  if (mx > 64) mx = 64;
  long long best = 1;
  double best_cost = 1e30;
  for (long long c = 1; c <= mx; ++c) {
    const long long rounds = ((long long)n * c + num_cus - 1) / num_cus;
    const double cost = (double)rounds / (double)c + 0.002 * (double)rounds;  // per-round set-up/fold
    if (cost < best_cost - 1e-12) { best_cost = cost; best = c; }
  }
  return best;
}

// What one call launches
struct HistPlan {
  int kernel;        // 16: the 16-bin instance, 32: 32 copies / 1024 threads, 2: packed two-per-CU, 8: eight copies / 256 threads, 0: one copy per wave
  bool half_steps;
  long long chunks;
  int mode;          // HC_*
};

HistPlan hist_plan(st_ctx* ctx, int n, int h, int w, int bins) {
  const long long nbytes = 3LL * h * w;
  const long long nvec = nbytes / 16;
  // ST_HIST_VARIANT selects the older kernels for A/B runs: 8 = eight copies per 256-thread
  // workgroup, 0 = one copy per wave.  Default: 32 lane-indexed copies, 1024 threads.
  static const int variant = getenv("ST_HIST_VARIANT") ? atoi(getenv("ST_HIST_VARIANT")) : 32;
  // the reference's own bin count (16) gets counters for byte >> 4 directly: 6 KB of LDS instead of 96 KB, 256-thread workgroups,
  // eight resident per CU -- clearing and folding shrink 16-fold and overlap with other workgroups' streaming
  // (ST_HIST16=0 keeps the general kernel)
  static const bool use16 = !(getenv("ST_HIST16") && atoi(getenv("ST_HIST16")) == 0);
  // ST_HIST_P2=1: the packed two-workgroups-per-CU kernel instead of the one-per-CU one with 96 KB of whole-dword counters.
  // Measured (round 4, 1080p, HIP events): 32 frames per launch +2 % on noise, -3 % on smooth / all-equal frames; 64 frames
  // +1..4 %; 256 frames +1..4 % -- inside the spread between boxes, so the default stays the kernel without a counter bound.
  static const bool use_p2 = getenv("ST_HIST_P2") && atoi(getenv("ST_HIST_P2")) == 1;
  // Half-steps (three vectors re-requested as soon as they are counted) shorten the drain after a workgroup's last load from
  // six vectors' counting to three: 37.5 -> 35.8 us for 32 frames per launch (8 pipelined steps per workgroup), nothing at 64
  // (16 steps), -1..4 % at 256 on noise (each load then has half a step to land).  Used for launches of at most 10 steps per
  // workgroup; ST_HIST_HALF=0 / 1 forces it off / on.
  static const int half_env = getenv("ST_HIST_HALF") ? atoi(getenv("ST_HIST_HALF")) : -1;
  // ST_HIST_COMMIT=atomic: memset + global atomics also where a frame has one workgroup (A/B)
  static const bool atomic_commit = getenv("ST_HIST_COMMIT") && std::string(getenv("ST_HIST_COMMIT")) == "atomic";
  HistPlan p;
  p.half_steps = false;
  p.mode = HC_ATOMIC;
  const int nl = n < 65535 ? n : 65535;   // frames per launch (grid.y)
  if (variant == 32 && bins == 16 && use16) {
    p.kernel = 16;
    p.chunks = ((long long)ctx->num_cus * 16 + n - 1) / n;
    long long max_chunks = (nvec + 3 * 256 * 4 - 1) / (3 * 256 * 4);
    if (p.chunks > max_chunks) p.chunks = max_chunks;
    if (p.chunks < 1) p.chunks = 1;
  } else if (variant == 32 && use_p2) {
    // two workgroups per CU: 2 * CUs slots; a chunk's half-counters must not overflow
    p.kernel = 2;
    p.chunks = chunks_for(2 * ctx->num_cus, nl, nvec, 1024);
    const long long need = (nvec + kP2MaxVec - 6 * 1024 - 1) / (kP2MaxVec - 6 * 1024);   // `per` is rounded up to 6 T vectors
    if (p.chunks < need) p.chunks = need;
    static const int force_chunks = getenv("ST_HIST_CHUNKS") ? atoi(getenv("ST_HIST_CHUNKS")) : 0;  // experiments
    if (force_chunks > need) p.chunks = force_chunks;
  } else if (variant == 32) {
    p.kernel = 32;
    p.chunks = chunks_for(ctx->num_cus, nl, nvec, 1024);
    static const int force_chunks = getenv("ST_HIST_CHUNKS") ? atoi(getenv("ST_HIST_CHUNKS")) : 0;  // experiments
    if (force_chunks > 0) p.chunks = force_chunks;
    const long long steps_per_wg = ((nvec + p.chunks - 1) / p.chunks + 6 * 1024 - 1) / (6 * 1024);
    p.half_steps = half_env == 1 || (half_env != 0 && steps_per_wg <= 10);
  } else {
    p.kernel = variant == 8 ? 8 : 0;
    p.chunks = ((long long)ctx->num_cus * 16 + n - 1) / n;
    long long max_chunks = (nvec + 3 * kThreads * 4 - 1) / (3 * kThreads * 4);  // >= 12 vectors per thread
    if (p.chunks > max_chunks) p.chunks = max_chunks;
    if (p.chunks < 1) p.chunks = 1;
  }
  if ((p.kernel == 16 || p.kernel == 32) && !atomic_commit && p.chunks == 1) p.mode = HC_DIRECT;
  return p;
}

int hist_launch(st_ctx* ctx, FrameSrc src, int n, int h, int w, int bins, int32_t* out_dev, const HistPlan& plan) {
  const long long nbytes = 3LL * h * w;
  const long long chunks = plan.chunks;
  const HistCommit hc{plan.mode};
  if (plan.mode == HC_ATOMIC) ST_HIP(ctx, hipMemsetAsync(out_dev, 0, sizeof(int32_t) * 3 * (size_t)bins * n, ctx->stream));
  // grid.y is limited to 65535 frames per launch
  for (int f0 = 0; f0 < n; f0 += 65535) {
    int nf = n - f0 < 65535 ? n - f0 : 65535;
    FrameSrc s = src;
    if (s.ptrs) s.ptrs += f0; else s.base += (size_t)f0 * s.stride;
    int32_t* o = out_dev + (size_t)f0 * 3 * bins;
    const dim3 grid((unsigned)chunks, (unsigned)nf);
    // timing events travel with the dispatch itself (st_time_dispatch): the figure is the kernel's own duration
    hipEvent_t e0, e1;
    ST_TRY(st_time_dispatch(ctx, ST_K_HIST, &e0, &e1));
    if (plan.kernel == 16)
      hipExtLaunchKernelGGL((k_hist_u8c3_v2<32, 256, 4>), grid, dim3(256), 0, ctx->stream, e0, e1, 0, s, nbytes, (int)chunks, bins, o, hc);
    else if (plan.kernel == 2)
      hipExtLaunchKernelGGL((k_hist_u8c3_p2<32, 1024>), grid, dim3(1024), 0, ctx->stream, e0, e1, 0, s, nbytes, (int)chunks, bins, o);
    else if (plan.kernel == 32 && plan.half_steps)
      hipExtLaunchKernelGGL((k_hist_u8c3_v2<32, 1024, 0, true>), grid, dim3(1024), 0, ctx->stream, e0, e1, 0, s, nbytes, (int)chunks, bins, o, hc);
    else if (plan.kernel == 32)
      hipExtLaunchKernelGGL((k_hist_u8c3_v2<32, 1024>), grid, dim3(1024), 0, ctx->stream, e0, e1, 0, s, nbytes, (int)chunks, bins, o, hc);
    else if (plan.kernel == 8)
      hipExtLaunchKernelGGL((k_hist_u8c3_v2<8, 256>), grid, dim3(256), 0, ctx->stream, e0, e1, 0, s, nbytes, (int)chunks, bins, o, hc);
    else
      hipExtLaunchKernelGGL(k_hist_u8c3, grid, dim3(kThreads), 0, ctx->stream, e0, e1, 0, s, nbytes, (int)chunks, bins, o);
    ST_HIP(ctx, hipGetLastError());
  }
  return ST_OK;
}

int hist_check(st_ctx* ctx, int n, int h, int w, int bins, const void* out) {
  if (n < 0 || h <= 0 || w <= 0 || bins < 1 || bins > 256 || (n > 0 && !out))
    return st_set_error(ctx, ST_ERR_INVALID, "histogram: bad arguments (n=%d h=%d w=%d bins=%d)", n, h, w, bins);
  return ST_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// ShotBoundaries on the device: the tail of Histogram -> ShotBoundaries.
// Replaces the body of shot_boundaries (/root/reference/scannertools/scannertools/shot_detection.py:12-28) for
// histograms that are already on the GPU: with the frames resident in HBM the Histogram kernel delivers 10 000 1080p
// frames in 14 ms, and the reference's host loop over 10 000 windows (29 ms even with its interior vectorised) is then
// two thirds of the pipeline.  The decisions must equal the reference's bit for bit -- a boundary is `diffs[i] -
// mean(win) > 2.5 std(win)` on float64 -- so the window sums are formed in EXACTLY numpy's order:
// np.add.reduce on a contiguous float64 vector is the pairwise summation of numpy/_core/src/umath/loops_utils.h.src
// (n < 8: running sum from 0; n <= 128: eight interleaved accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
// and the n mod 8 tail added in order; larger n: split at n/2 rounded down to a multiple of 8), np.mean = that sum /
// n, np.std = sqrt(pairwise((x - mean)^2) / n) with x - mean and its square rounded separately
// (numpy/_core/_methods.py: _mean, _var, _std).  tests/test_shots_gpu.py checks the emulation against numpy itself,
// the flags against the golden lists produced by importing the reference, and against the host op on random streams.
// One thread per frame (a window is <= 2 W values read three times, L1/L2-resident): 10 000 frames in ~40 us.
// ---------------------------------------------------------------------------------------------------------------
// a block of at most 128 values: numpy's unrolled loop
template <class F>
__device__ __forceinline__ double np_pairwise_leaf(const F& f, int lo, int n) {
  if (n < 8) {
    double r = 0.;
    for (int i = 0; i < n; ++i) r += f(lo + i);
    return r;
  }
  double r0 = f(lo), r1 = f(lo + 1), r2 = f(lo + 2), r3 = f(lo + 3), r4 = f(lo + 4), r5 = f(lo + 5), r6 = f(lo + 6), r7 = f(lo + 7);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
    r0 += f(lo + i); r1 += f(lo + i + 1); r2 += f(lo + i + 2); r3 += f(lo + i + 3);
    r4 += f(lo + i + 4); r5 += f(lo + i + 5); r6 += f(lo + i + 6); r7 += f(lo + i + 7);
  }
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < n; ++i) res += f(lo + i);
  return res;
}

// numpy's recursion (n > 128: pairwise(left half rounded down to a multiple of 8) + pairwise(rest)) evaluated with an
// explicit stack in the same post-order, so that the depth a thread needs is a compile-time bound instead of a device
// call stack: a range halves down to <= 128 values in at most 24 levels for n < 2^31.
template <class F>
__device__ double np_pairwise(const F& f, int lo, int n) {
  if (n <= 128) return np_pairwise_leaf(f, lo, n);
  constexpr int kDepth = 32;
  int s_lo[kDepth], s_n[kDepth], s_state[kDepth];
  double s_left[kDepth];
  int sp = 0;
  s_lo[0] = lo; s_n[0] = n; s_state[0] = 0; s_left[0] = 0.;
  double result = 0.;
  while (sp >= 0) {
    const int cn = s_n[sp], clo = s_lo[sp];
    if (cn <= 128) {
      result = np_pairwise_leaf(f, clo, cn);
      --sp;
      continue;
    }
    int n2 = cn / 2;
    n2 -= n2 % 8;
    if (s_state[sp] == 0) {          // descend into the left part
      s_state[sp] = 1;
      ++sp;
      s_lo[sp] = clo; s_n[sp] = n2; s_state[sp] = 0;
    } else if (s_state[sp] == 1) {   // left part done: keep it, descend into the right part
      s_left[sp] = result;
      s_state[sp] = 2;
      ++sp;
      s_lo[sp] = clo + n2; s_n[sp] = cn - n2; s_state[sp] = 0;
    } else {                         // both done
      result = s_left[sp] + result;
      --sp;
    }
  }
  return result;
}

// diffs[0] = 0; diffs[i] = mean over the 3 channels of max_b |h[i-1][c][b] - h[i][c][b]| (shot_detection.py:14-18:
// scipy's chebyshev on the integer histograms, np.mean of the three integers: an exact sum, one division)
__global__ __launch_bounds__(256) void k_shot_diffs(const int32_t* __restrict__ hist, int n, int bins, double* __restrict__ diffs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (i == 0) { diffs[0] = 0.; return; }
  const int32_t* a = hist + (size_t)(i - 1) * 3 * bins;
  const int32_t* b = a + 3 * bins;
  long long tot = 0;
  for (int c = 0; c < 3; ++c) {
    long long mx = 0;
    for (int k = 0; k < bins; ++k) {
      long long d = (long long)a[c * bins + k] - (long long)b[c * bins + k];
      d = d < 0 ? -d : d;
      mx = d > mx ? d : mx;
    }
    tot += mx;
  }
  diffs[i] = (double)tot / 3.0;
}

__global__ __launch_bounds__(64) void k_shot_outliers(const double* __restrict__ diffs, int n, int W, double kstd, uint8_t* __restrict__ flags) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  if (i == 0) { flags[0] = 0; return; }
  const int lo = i - W > 0 ? i - W : 0, hi = i + W < n ? i + W : n, m = hi - lo;
  auto val = [&](int j) { return diffs[j]; };
  const double mean = __ddiv_rn(np_pairwise(val, lo, m), (double)m);
  auto sq = [&](int j) { const double x = diffs[j] - mean; return x * x; };
  const double sd = __dsqrt_rn(__ddiv_rn(np_pairwise(sq, lo, m), (double)m));
  flags[i] = (diffs[i] - mean > kstd * sd) ? 1 : 0;
}

}  // namespace

ST_EXPORT int st_shot_boundaries(st_ctx* ctx, const int32_t* hist_dev, int n, int bins, int window, double k_std, uint8_t* flags_dev,
                                 double* diffs_dev) {
  ST_TRY(st_enter(ctx));
  if (n < 0 || bins < 1 || bins > 65536 || window < 1 || (n > 0 && (!hist_dev || !flags_dev)))
    return st_set_error(ctx, ST_ERR_INVALID, "shot boundaries: bad arguments (n=%d bins=%d window=%d)", n, bins, window);
  if (n == 0) return ST_OK;
  // a window of more than n frames on either side is the whole stream for every frame (and i + window must stay an int)
  if (window > n) window = n;
  double* d = diffs_dev;
  if (!d) {
    ST_TRY(st_ws_reserve(ctx, st_align_up(sizeof(double) * (size_t)n)));
    d = (double*)st_ws_alloc(ctx, sizeof(double) * (size_t)n);
  }
  hipLaunchKernelGGL(k_shot_diffs, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, hist_dev, n, bins, d);
  ST_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(k_shot_outliers, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, d, n, window, k_std, flags_dev);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

ST_EXPORT int st_hist_u8c3_batch(st_ctx* ctx, const uint8_t* const* frames_dev, int n, int h, int w, int bins,
                                 int32_t* out_dev) {
  ST_TRY(st_enter(ctx));
  ST_TRY(hist_check(ctx, n, h, w, bins, out_dev));
  if (n == 0) return ST_OK;
  if (!frames_dev) return st_set_error(ctx, ST_ERR_INVALID, "histogram: null frame table");
  for (int i = 0; i < n; ++i)
    if (!frames_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "histogram: frame %d is null", i);
  const HistPlan plan = hist_plan(ctx, n, h, w, bins);
  ST_TRY(st_ws_reserve(ctx, st_align_up(sizeof(void*) * (size_t)n)));
  const uint8_t** table = (const uint8_t**)st_ws_alloc(ctx, sizeof(void*) * (size_t)n);
  ST_HIP(ctx, hipMemcpyAsync(table, frames_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  FrameSrc src{table, nullptr, 0};
  return hist_launch(ctx, src, n, h, w, bins, out_dev, plan);
}

ST_EXPORT int st_hist_u8c3_strided(st_ctx* ctx, const uint8_t* base_dev, size_t frame_stride_bytes, int n, int h,
                                   int w, int bins, int32_t* out_dev) {
  ST_TRY(st_enter(ctx));
  ST_TRY(hist_check(ctx, n, h, w, bins, out_dev));
  if (n == 0) return ST_OK;
  if (!base_dev || frame_stride_bytes < (size_t)3 * h * w)
    return st_set_error(ctx, ST_ERR_INVALID, "histogram: bad base/stride");
  FrameSrc src{nullptr, base_dev, frame_stride_bytes};
  const HistPlan plan = hist_plan(ctx, n, h, w, bins);
  return hist_launch(ctx, src, n, h, w, bins, out_dev, plan);
}
