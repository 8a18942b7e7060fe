// Consumers of the flow field: FlowHistogram and DrawFlow.
//
// FlowHistogram replaces FlowHistogramKernelCPU::execute
// (/root/reference/scannertools/scannertools/old/cpp_ops/flow_histogram_kernel_cpu.cpp:26-57):
// cv::split + cv::cartToPolar(deg) + two 64-bin cv::calcHist passes + convertTo(CV_32S) become
// ONE pass over the flow frame (8 B/px read, 512 B written per frame).
// DrawFlow replaces the numpy body of /root/reference/scannertools/scannertools/vis.py:8-12:
// a max-reduction pass (8 B/px) and a render pass (frame 3 + flow 8 in, 6 out per px).
//
// Arithmetic follows oracle/oracle.c (scalar non-FMA OpenCV / numpy float32 semantics): float
// sqrt and divide are the correctly rounded ones (hipcc default), -ffp-contract=off keeps the
// polynomial un-fused, bin indices are computed in double as cv::calcHist does.
#include <cfloat>
#include <climits>

#include "st_internal.h"

namespace {

constexpr int kT = 256;
constexpr int kCopies = 32;   // lane-indexed counter copies: a lane always hits its own LDS bank
constexpr int kFlowBins = 64; // flow_histogram_kernel_cpu.cpp:9

struct FlowSrc {
  const float* const* ptrs;  // device table of frame pointers, or null
  const float* base;         // strided stream
  size_t stride;             // bytes
  __device__ const float* frame(int i) const {
    return ptrs ? st_gl(ptrs[i]) : reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (size_t)i * stride);
  }
};

// cv::fastAtan2's polynomial in degrees (core/src/mathfuncs_core.simd.hpp atan_f32)
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
  const float ax = fabsf(x), ay = fabsf(y);
  const bool xs = ax >= ay;
  const float num = xs ? ay : ax, den = xs ? ax : ay;
  const float c = num / (den + (float)DBL_EPSILON);
  const float c2 = c * c;
  float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  if (!xs) a = 90.f - a;
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

__device__ __forceinline__ void lds_inc(unsigned* p) {
  __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// one flow vector -> its two counters (cv::calcHist uniform float: idx = floor(double(v)*a + b),
// counted iff 0 <= idx < 64)
__device__ __forceinline__ void count_px(unsigned* sh, unsigned copy, float fx, float fy) {
  const float mag = sqrtf(fx * fx + fy * fy);
  const float deg = fast_atan2_deg(fy, fx) * 1.f;
  const double a_mag = kFlowBins / (64.0 - 0.0), a_deg = kFlowBins / (360.0 - 0.0);
  const double dm = (double)mag * a_mag + (-a_mag * 0.0);
  const double dd = (double)deg * a_deg + (-a_deg * 0.0);
  if (dm >= 0.0 && dm < (double)kFlowBins) lds_inc(sh + (unsigned)(int)dm * kCopies + copy);
  if (dd >= 0.0 && dd < (double)kFlowBins) lds_inc(sh + (kFlowBins + (unsigned)(int)dd) * kCopies + copy);
}

__global__ __launch_bounds__(kT) void k_flow_hist(FlowSrc src, long long npx, int chunks, int32_t* __restrict__ out) {
  __shared__ unsigned sh[2 * kFlowBins * kCopies];
  const int tid = threadIdx.x;
  const int frame = blockIdx.y, chunk = blockIdx.x;
  const float* p = src.frame(frame);
  for (int i = tid; i < 2 * kFlowBins * kCopies; i += kT) sh[i] = 0;
  __syncthreads();
  const unsigned copy = tid & (kCopies - 1);

  // [0, head) and [tail, npx) are the pixels outside the 16-B aligned body, read as float2
  long long head = (long long)(((16 - ((uintptr_t)p & 15)) & 15) / 8);  // flow frames are 8-B aligned
  if (head > npx) head = npx;
  const long long nvec = (npx - head) >> 1;  // float4 = 2 px
  const long long tail = head + (nvec << 1);
  // the flow field is read once: non-temporal loads (as in the Histogram kernel)
  typedef float f4nt __attribute__((ext_vector_type(4)));
  const f4nt* vq = reinterpret_cast<const f4nt*>(p + 2 * head);
  auto ld = [&](long long j) { const f4nt v = __builtin_nontemporal_load(vq + j); return make_float4(v.x, v.y, v.z, v.w); };
  const long long per = (nvec + chunks - 1) / chunks;
  const long long v0 = (long long)chunk * per;
  long long v1 = v0 + per;
  if (v1 > nvec) v1 = nvec;

  long long i = v0 + tid;
  for (; i + 3 * kT < v1; i += 4 * kT) {
    const float4 a = ld(i), b = ld(i + kT), c = ld(i + 2 * kT), d = ld(i + 3 * kT);
    count_px(sh, copy, a.x, a.y); count_px(sh, copy, a.z, a.w);
    count_px(sh, copy, b.x, b.y); count_px(sh, copy, b.z, b.w);
    count_px(sh, copy, c.x, c.y); count_px(sh, copy, c.z, c.w);
    count_px(sh, copy, d.x, d.y); count_px(sh, copy, d.z, d.w);
  }
  for (; i < v1; i += kT) {
    const float4 a = ld(i);
    count_px(sh, copy, a.x, a.y); count_px(sh, copy, a.z, a.w);
  }
  if (chunk == 0) {
    const float2* q = reinterpret_cast<const float2*>(p);
    for (long long b = tid; b < head; b += kT) count_px(sh, copy, q[b].x, q[b].y);
    for (long long b = tail + tid; b < npx; b += kT) count_px(sh, copy, q[b].x, q[b].y);
  }
  __syncthreads();
  if (tid < 2 * kFlowBins) {
    unsigned s = 0;
#pragma unroll
    for (int c = 0; c < kCopies; ++c) s += sh[tid * kCopies + ((c + tid) & (kCopies - 1))];
    if (s) atomicAdd(reinterpret_cast<unsigned*>(out) + (size_t)frame * 2 * kFlowBins + tid, s);
  }
}

// ---- DrawFlow ---------------------------------------------------------------------------------

// order-preserving float -> int key (for atomicMax on the per-frame maximum)
__device__ __forceinline__ int float_key(float v) {
  const int b = __float_as_int(v);
  return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float key_float(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

// stats[2*frame] = key of max((fx+fy)/2), stats[2*frame+1] = 1 if any value is NaN (np.max propagates it)
__global__ __launch_bounds__(kT) void k_flow_avg_max(FlowSrc src, long long npx, int chunks, int* __restrict__ stats) {
  __shared__ int smax[kT / 64];
  __shared__ int snan[kT / 64];
  const int tid = threadIdx.x, frame = blockIdx.y, chunk = blockIdx.x;
  const float2* q = reinterpret_cast<const float2*>(src.frame(frame));
  const long long per = (npx + chunks - 1) / chunks;
  const long long i0 = (long long)chunk * per;
  long long i1 = i0 + per;
  if (i1 > npx) i1 = npx;
  float m = -INFINITY;
  int nan = 0;
  for (long long i = i0 + tid; i < i1; i += kT) {
    const float2 f = q[i];
    const float v = (f.x + f.y) / 2.f;
    nan |= (v != v);
    m = v > m ? v : m;
  }
  int key = float_key(m);
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(key, off);
    key = o > key ? o : key;
    nan |= __shfl_xor(nan, off);
  }
  if ((tid & 63) == 0) { smax[tid >> 6] = key; snan[tid >> 6] = nan; }
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < kT / 64; ++k) { key = smax[k] > key ? smax[k] : key; nan |= snan[k]; }
    atomicMax(stats + 2 * frame, key);
    if (nan) atomicOr(stats + 2 * frame + 1, 1);
  }
}

// numpy's float32 -> uint8 C cast: truncate toward zero to int32 (NaN / out of range -> INT_MIN),
// keep the low byte
__device__ __forceinline__ unsigned vis_byte(float fx, float fy, float m) {
  const float v = (fx + fy) / 2.f;
  float q = v / m;
  q = q > 1.0f ? 1.0f : q;  // np.clip(., None, 1): NaN stays NaN
  q = q * 255.f;
  const int t = (q >= -2147483648.f && q < 2147483648.f) ? (int)q : INT_MIN;
  return (unsigned)t & 0xffu;
}

struct DrawArgs {
  const uint8_t* const* frames;
  FlowSrc flows;
  uint8_t* const* outs;
  const int* stats;
  int h, w;
};

// thread = 4 pixels of one row (w % 4 == 0, so every row of frame and output starts 4-B aligned)
__global__ __launch_bounds__(kT) void k_draw_flow4(DrawArgs a) {
  const int frame = blockIdx.z, y = blockIdx.y;
  const int x4 = blockIdx.x * kT + threadIdx.x;
  if (x4 * 4 >= a.w) return;
  const int nan = a.stats[2 * frame + 1];
  const float m = nan ? NAN : key_float(a.stats[2 * frame]);
  const uint8_t* fr = st_gl(a.frames[frame]) + ((size_t)y * a.w + (size_t)x4 * 4) * 3;
  const float* fl = a.flows.frame(frame) + ((size_t)y * a.w + (size_t)x4 * 4) * 2;
  uint8_t* o = st_gl(a.outs[frame]) + (size_t)y * a.w * 6 + (size_t)x4 * 12;
  const uint32_t* fr32 = reinterpret_cast<const uint32_t*>(fr);
  const uint32_t c0 = fr32[0], c1 = fr32[1], c2 = fr32[2];
  const float4 f0 = reinterpret_cast<const float4*>(fl)[0], f1 = reinterpret_cast<const float4*>(fl)[1];
  const unsigned b0 = vis_byte(f0.x, f0.y, m), b1 = vis_byte(f0.z, f0.w, m);
  const unsigned b2 = vis_byte(f1.x, f1.y, m), b3 = vis_byte(f1.z, f1.w, m);
  uint32_t* ol = reinterpret_cast<uint32_t*>(o);
  ol[0] = c0; ol[1] = c1; ol[2] = c2;
  uint32_t* orr = reinterpret_cast<uint32_t*>(o + (size_t)a.w * 3);
  orr[0] = b0 * 0x010101u | (b1 << 24);
  orr[1] = b1 * 0x0101u | (b2 << 16) | (b2 << 24);
  orr[2] = b2 | (b3 * 0x01010100u);
}

// any width / alignment: thread = pixel
__global__ __launch_bounds__(kT) void k_draw_flow1(DrawArgs a) {
  const int frame = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * kT + threadIdx.x;
  if (x >= a.w) return;
  const int nan = a.stats[2 * frame + 1];
  const float m = nan ? NAN : key_float(a.stats[2 * frame]);
  const uint8_t* fr = st_gl(a.frames[frame]) + ((size_t)y * a.w + x) * 3;
  const float* fl = a.flows.frame(frame) + ((size_t)y * a.w + x) * 2;
  uint8_t* o = st_gl(a.outs[frame]) + (size_t)y * a.w * 6;
  const uint8_t b = (uint8_t)vis_byte(fl[0], fl[1], m);
  o[x * 3 + 0] = fr[0]; o[x * 3 + 1] = fr[1]; o[x * 3 + 2] = fr[2];
  uint8_t* r = o + (size_t)a.w * 3 + (size_t)x * 3;
  r[0] = b; r[1] = b; r[2] = b;
}

__global__ void k_init_stats(int* stats, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { stats[2 * i] = INT_MIN; stats[2 * i + 1] = 0; }
}

long long chunks_for_grid(int num_cus, int n, long long units_per_frame, long long min_units) {
  long long chunks = ((long long)num_cus * 16 + n - 1) / n;
  long long mx = units_per_frame / min_units;
  if (chunks > mx) chunks = mx;
  return chunks < 1 ? 1 : chunks;
}

int flow_hist_launch(st_ctx* ctx, FlowSrc src, int n, int h, int w, int32_t* out_dev) {
  const long long npx = (long long)h * w;
  const long long chunks = chunks_for_grid(ctx->num_cus, n, npx / 2, 8 * kT);
  ST_HIP(ctx, hipMemsetAsync(out_dev, 0, sizeof(int32_t) * 2 * kFlowBins * (size_t)n, ctx->stream));
  for (int f0 = 0; f0 < n; f0 += 65535) {
    const int nf = n - f0 < 65535 ? n - f0 : 65535;
    FlowSrc s = src;
    if (s.ptrs) s.ptrs += f0; else s.base = reinterpret_cast<const float*>(reinterpret_cast<const char*>(s.base) + (size_t)f0 * s.stride);
    st_timed t(ctx, ST_K_FLOW_HIST);
    hipLaunchKernelGGL(k_flow_hist, dim3((unsigned)chunks, (unsigned)nf), dim3(kT), 0, ctx->stream, s, npx, (int)chunks,
                       out_dev + (size_t)f0 * 2 * kFlowBins);
    ST_HIP(ctx, hipGetLastError());
  }
  return ST_OK;
}

int flow_check(st_ctx* ctx, const char* what, int n, int h, int w) {
  if (n < 0 || h <= 0 || w <= 0 || (long long)h * w > 200000000LL)
    return st_set_error(ctx, ST_ERR_INVALID, "%s: bad arguments (n=%d h=%d w=%d)", what, n, h, w);
  return ST_OK;
}

}  // namespace

ST_EXPORT int st_flow_hist_batch(st_ctx* ctx, const float* const* flows_dev, int n, int h, int w, int32_t* out_dev) {
  ST_TRY(st_enter(ctx));
  ST_TRY(flow_check(ctx, "flow histogram", n, h, w));
  if (n == 0) return ST_OK;
  if (!flows_dev || !out_dev) return st_set_error(ctx, ST_ERR_INVALID, "flow histogram: null argument");
  for (int i = 0; i < n; ++i) {
    if (!flows_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "flow histogram: flow %d is null", i);
    if ((uintptr_t)flows_dev[i] & 7) return st_set_error(ctx, ST_ERR_INVALID, "flow histogram: flow %d is not 8-byte aligned", i);
  }
  ST_TRY(st_ws_reserve(ctx, st_align_up(sizeof(void*) * (size_t)n)));
  const float** table = (const float**)st_ws_alloc(ctx, sizeof(void*) * (size_t)n);
  ST_HIP(ctx, hipMemcpyAsync(table, flows_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  FlowSrc src{table, nullptr, 0};
  return flow_hist_launch(ctx, src, n, h, w, out_dev);
}

ST_EXPORT int st_flow_hist_strided(st_ctx* ctx, const float* base_dev, size_t frame_stride_bytes, int n, int h, int w,
                                   int32_t* out_dev) {
  ST_TRY(st_enter(ctx));
  ST_TRY(flow_check(ctx, "flow histogram", n, h, w));
  if (n == 0) return ST_OK;
  if (!base_dev || !out_dev || frame_stride_bytes < (size_t)8 * h * w || (frame_stride_bytes & 7) || ((uintptr_t)base_dev & 7))
    return st_set_error(ctx, ST_ERR_INVALID, "flow histogram: bad base/stride");
  FlowSrc src{nullptr, base_dev, frame_stride_bytes};
  return flow_hist_launch(ctx, src, n, h, w, out_dev);
}

ST_EXPORT int st_draw_flow_batch(st_ctx* ctx, const uint8_t* const* frames_dev, const float* const* flows_dev, int n, int h,
                                 int w, uint8_t* const* out_dev) {
  ST_TRY(st_enter(ctx));
  ST_TRY(flow_check(ctx, "draw flow", n, h, w));
  if (n == 0) return ST_OK;
  if (!frames_dev || !flows_dev || !out_dev) return st_set_error(ctx, ST_ERR_INVALID, "draw flow: null argument");
  bool aligned = (w % 4) == 0;
  for (int i = 0; i < n; ++i) {
    if (!frames_dev[i] || !flows_dev[i] || !out_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "draw flow: row %d is null", i);
    if ((uintptr_t)flows_dev[i] & 7) return st_set_error(ctx, ST_ERR_INVALID, "draw flow: flow %d is not 8-byte aligned", i);
    if (((uintptr_t)frames_dev[i] & 3) || ((uintptr_t)out_dev[i] & 3) || ((uintptr_t)flows_dev[i] & 15)) aligned = false;
  }
  const size_t tb = st_align_up(sizeof(void*) * (size_t)n);
  ST_TRY(st_ws_reserve(ctx, 3 * tb + st_align_up(sizeof(int) * 2 * (size_t)n)));
  const uint8_t** d_frames = (const uint8_t**)st_ws_alloc(ctx, tb);
  const float** d_flows = (const float**)st_ws_alloc(ctx, tb);
  uint8_t** d_outs = (uint8_t**)st_ws_alloc(ctx, tb);
  int* stats = (int*)st_ws_alloc(ctx, sizeof(int) * 2 * (size_t)n);
  ST_HIP(ctx, hipMemcpyAsync(d_frames, frames_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  ST_HIP(ctx, hipMemcpyAsync(d_flows, flows_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  ST_HIP(ctx, hipMemcpyAsync(d_outs, out_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  const long long npx = (long long)h * w;
  st_timed t(ctx, ST_K_DRAW_FLOW);
  for (int f0 = 0; f0 < n; f0 += 65535) {
    const int nf = n - f0 < 65535 ? n - f0 : 65535;
    hipLaunchKernelGGL(k_init_stats, dim3((nf + 255) / 256), dim3(256), 0, ctx->stream, stats + 2 * (size_t)f0, nf);
    FlowSrc src{d_flows + f0, nullptr, 0};
    const long long chunks = chunks_for_grid(ctx->num_cus, nf, npx, 4 * kT);
    hipLaunchKernelGGL(k_flow_avg_max, dim3((unsigned)chunks, (unsigned)nf), dim3(kT), 0, ctx->stream, src, npx, (int)chunks,
                       stats + 2 * (size_t)f0);
    DrawArgs a{d_frames + f0, src, d_outs + f0, stats + 2 * (size_t)f0, h, w};
    if (h > 65535) return st_set_error(ctx, ST_ERR_INVALID, "draw flow: height %d exceeds the grid limit", h);
    if (aligned)
      hipLaunchKernelGGL(k_draw_flow4, dim3((w / 4 + kT - 1) / kT, h, nf), dim3(kT), 0, ctx->stream, a);
    else
      hipLaunchKernelGGL(k_draw_flow1, dim3((w + kT - 1) / kT, h, nf), dim3(kT), 0, ctx->stream, a);
    ST_HIP(ctx, hipGetLastError());
  }
  return ST_OK;
}
