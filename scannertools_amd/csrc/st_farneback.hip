// Farneback dense optical flow for gfx950.
//
// Replaces the arithmetic of OpticalFlowKernelCPU::execute
// (/root/reference/scannertools/scannertools_cpp/imgproc/optical_flow_kernel_cpu.cpp:36-41):
//   cv::cvtColor(BGR2GRAY) x2  ->  k_gray4 / k_gray
//   cv::FarnebackOpticalFlow::calc:
//     convertTo(F32) + GaussianBlur + resize        ->  k_pyr_fused      (all four levels in one pass over
//                                                       the gray frame; k_pyr0 / k_pyr_dec / k_pyr per
//                                                       level for other geometries)
//     FarnebackPolyExp                              ->  k_polyexp        (row-marching, LDS exchange)
//     resize(prevFlow)*2 + FarnebackUpdateMatrices
//       + FarnebackUpdateFlow_Blur                  ->  k_flow_iter3 / k_flow_iter_tile (one launch per
//                                                       iteration: M is recomputed on the fly, never stored)
//     the same stages unfused                       ->  k_update_matrices, k_blur_update(_v2)
//                                                       (window sizes other than 15, stage tests)
// All kernels are batched over frames / pairs through blockIdx.z.  The operand order and the
// accumulator types (float vs double) of every expression follow the OpenCV scalar code so
// that, built with -ffp-contract=off, every stage except the running-sum box filter is
// bit-identical to the CPU restatement in oracle/oracle.c.
//
// Data layout in HBM (dense, per frame or per pair):
//   gray  u8 (h,w)            I_k  f32 (lh,lw)
//   R_k   f32, per frame lh*lw float4 [d/dy, d/dx, yy, xx] then lh*lw float [xy]: one 16-B and one
//         4-B load per pixel, coalesced across a wave (also for the bilinear gather of R1 when
//         the flow is smooth); see the note above update_matrices_px
//   M     f32 planar (5,lh,lw): [G11, G12, G22, h1, h2] -- unfused path only (ping-pong per iteration)
//   flow  f32 (lh,lw,2) interleaved (u,v): the op's output format
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "st_internal.h"

namespace {

constexpr int kMaxTaps = 32;   // Gaussian pyramid kernel taps (reference needs <= 19)
constexpr int kMaxPolyN = 7;

// ---------------------------------------------------------------------------------------------
// host-side replicas of the scalar parameter derivations in optflowgf.cpp / smooth.cpp
// ---------------------------------------------------------------------------------------------
inline int cv_round(double v) { return (int)lrint(v); }

struct LevelGeom {
  int lh, lw, ksize;
  double sigma, scale;
};

int fb_levels(int h, int w, const st_fb_params& p) {
  int k;
  double scale = 1;
  for (k = 0; k < p.num_levels; ++k) {
    scale *= p.pyr_scale;
    if (w * scale < 32 || h * scale < 32) break;
  }
  return k;
}

LevelGeom fb_level_geom(int h, int w, const st_fb_params& p, int k) {
  LevelGeom g;
  double scale = 1;
  for (int i = 0; i < k; ++i) scale *= p.pyr_scale;
  g.scale = scale;
  g.sigma = (1. / scale - 1) * 0.5;
  int sz = cv_round(g.sigma * 5) | 1;
  g.ksize = sz > 3 ? sz : 3;
  g.lw = cv_round(w * scale);
  g.lh = cv_round(h * scale);
  return g;
}

// cv::getGaussianKernel(n, sigma, CV_32F)
void gaussian_kernel(int n, double sigma, float* k) {
  if (n == 3 && sigma <= 0) { k[0] = 0.25f; k[1] = 0.5f; k[2] = 0.25f; return; }
  if (n == 1 && sigma <= 0) { k[0] = 1.f; return; }
  if (n == 5 && sigma <= 0) { k[0] = 0.0625f; k[1] = 0.25f; k[2] = 0.375f; k[3] = 0.25f; k[4] = 0.0625f; return; }
  if (n == 7 && sigma <= 0) {
    static const float t[7] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
    memcpy(k, t, sizeof(t));
    return;
  }
  double sx = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  double scale2x = -0.5 / (sx * sx);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    double x = i - (n - 1) * 0.5;
    k[i] = (float)std::exp(scale2x * x * x);
    sum += k[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) k[i] = (float)(k[i] * sum);
}

struct PolyCoef {
  float g[kMaxPolyN + 1], xg[kMaxPolyN + 1], xxg[kMaxPolyN + 1];  // taps 0..n (symmetric / antisymmetric)
  double ig11, ig03, ig33, ig55;
};

void invert6(double A[6][6], double inv[6][6]) {
  double a[6][12];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) { a[i][j] = A[i][j]; a[i][6 + j] = (i == j); }
  for (int c = 0; c < 6; ++c) {
    int p = c;
    for (int r = c + 1; r < 6; ++r) if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
    if (p != c) for (int j = 0; j < 12; ++j) { double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
    double d = 1. / a[c][c];
    for (int j = 0; j < 12; ++j) a[c][j] *= d;
    for (int r = 0; r < 6; ++r) if (r != c) {
      double f = a[r][c];
      if (f != 0) for (int j = 0; j < 12; ++j) a[r][j] -= f * a[c][j];
    }
  }
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) inv[i][j] = a[i][6 + j];
}

// FarnebackPrepareGaussian
void poly_prepare(int n, double sigma, PolyCoef* pc) {
  float gb[2 * kMaxPolyN + 1], xgb[2 * kMaxPolyN + 1], xxgb[2 * kMaxPolyN + 1];
  float *g = gb + n, *xg = xgb + n, *xxg = xxgb + n;
  if (sigma < 1.1920929e-07) sigma = n * 0.3;
  double s = 0.;
  for (int x = -n; x <= n; ++x) {
    g[x] = (float)std::exp(-x * x / (2 * sigma * sigma));
    s += g[x];
  }
  s = 1. / s;
  for (int x = -n; x <= n; ++x) {
    g[x] = (float)(g[x] * s);
    xg[x] = (float)(x * g[x]);
    xxg[x] = (float)(x * x * g[x]);
  }
  double G[6][6];
  memset(G, 0, sizeof(G));
  for (int y = -n; y <= n; ++y)
    for (int x = -n; x <= n; ++x) {
      G[0][0] += g[y] * g[x];
      G[1][1] += g[y] * g[x] * x * x;
      G[3][3] += g[y] * g[x] * x * x * x * x;
      G[5][5] += g[y] * g[x] * x * x * y * y;
    }
  G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
  G[4][4] = G[3][3];
  G[3][4] = G[4][3] = G[5][5];
  double inv[6][6];
  invert6(G, inv);
  memset(pc, 0, sizeof(*pc));
  for (int k = 0; k <= n; ++k) { pc->g[k] = g[k]; pc->xg[k] = xg[k]; pc->xxg[k] = xxg[k]; }
  pc->ig11 = inv[1][1]; pc->ig03 = inv[0][3]; pc->ig33 = inv[3][3]; pc->ig55 = inv[5][5];
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int d_reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}
__device__ __forceinline__ int d_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// 1/x for the 2x2 solve of the fused iteration: hardware estimate + two Newton steps (6
// instructions, within an ulp or two of the IEEE quotient) instead of the ~35-instruction correctly
// rounded division.  x = det + 1e-3 > 0.  The box-filter stage is compared under a tolerance (its
// running sums already associate differently from the reference), so the last bit of the double
// reciprocal is not part of the contract; the stage-level kernels keep the exact division.
__device__ __forceinline__ double d_rcp_pos(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}

// ---------------------------------------------------------------------------------------------
// A2: 8-bit luma with OpenCV's BGR table applied to RGB bytes (reference quirk).
// ---------------------------------------------------------------------------------------------
struct GrayArgs {
  const uint8_t* const* frames;  // device table (rgb)
  uint8_t* gray;                 // n x npix
  int npix;
  int cb, cg, cr, rnd, shift;
};

__global__ __launch_bounds__(256) void k_gray(GrayArgs a) {
  const uint8_t* __restrict__ src = st_gl(a.frames[blockIdx.y]);
  uint8_t* __restrict__ dst = a.gray + (size_t)blockIdx.y * a.npix;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.npix; i += gridDim.x * blockDim.x) {
    int c0 = src[3 * (size_t)i], c1 = src[3 * (size_t)i + 1], c2 = src[3 * (size_t)i + 2];
    dst[i] = (uint8_t)((c0 * a.cb + c1 * a.cg + c2 * a.cr + a.rnd) >> a.shift);
  }
}

// Four pixels per thread: 12 source bytes as three aligned dwords, one dword store.  Needs
// 4-byte aligned frames and npix % 4 == 0 (checked on the host); the byte kernel covers the rest.
__global__ __launch_bounds__(256) void k_gray4(GrayArgs a) {
  const unsigned* __restrict__ src = reinterpret_cast<const unsigned*>(st_gl(a.frames[blockIdx.y]));
  unsigned* __restrict__ dst = reinterpret_cast<unsigned*>(a.gray + (size_t)blockIdx.y * a.npix);
  const int n4 = a.npix >> 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    typedef unsigned u3nt __attribute__((ext_vector_type(3), aligned(4)));
    const u3nt wv = __builtin_nontemporal_load(reinterpret_cast<const u3nt*>(src + 3 * (size_t)i));  // frame bytes are read once (-3 %)
    const unsigned w0 = wv.x, w1 = wv.y, w2 = wv.z;
    // bytes: w0 = r0 g0 b0 r1 | w1 = g1 b1 r2 g2 | w2 = b2 r3 g3 b3   (little endian, "r" = byte 0 of a pixel)
    const int g0 = (int)((w0 & 0xff) * a.cb + ((w0 >> 8) & 0xff) * a.cg + ((w0 >> 16) & 0xff) * a.cr + a.rnd) >> a.shift;
    const int g1 = (int)((w0 >> 24) * a.cb + (w1 & 0xff) * a.cg + ((w1 >> 8) & 0xff) * a.cr + a.rnd) >> a.shift;
    const int g2 = (int)(((w1 >> 16) & 0xff) * a.cb + (w1 >> 24) * a.cg + (w2 & 0xff) * a.cr + a.rnd) >> a.shift;
    const int g3 = (int)(((w2 >> 8) & 0xff) * a.cb + ((w2 >> 16) & 0xff) * a.cg + (w2 >> 24) * a.cr + a.rnd) >> a.shift;
    dst[i] = (unsigned)g0 | ((unsigned)g1 << 8) | ((unsigned)g2 << 16) | ((unsigned)g3 << 24);
  }
}

// The three pointer / index tables of a pass (frames, pair indices, output frames).  Large batches upload them with three
// copies; a call of a few pairs is a chain of short dependent launches in which three pageable host-to-device copies cost
// 5 us each on the stream, so there the values travel in the ARGUMENTS of the first kernel of the pass (k_gray_tab: the luma
// conversion reads its frame pointers straight from its arguments and its first workgroup writes the tables the later
// kernels read) -- no copy and no extra launch.
constexpr int kTabMax = 40;
struct TablesArg {
  const uint8_t** d_frames;
  float** d_outs;
  int* d_pairs;
  int nf, npairs;
  const uint8_t* frames[kTabMax];
  float* outs[kTabMax];
  int pairs[2 * kTabMax];
};

__device__ __forceinline__ void gray4_body(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int n4, const GrayArgs& a) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    typedef unsigned u3nt __attribute__((ext_vector_type(3), aligned(4)));
    const u3nt wv = __builtin_nontemporal_load(reinterpret_cast<const u3nt*>(src + 3 * (size_t)i));
    const unsigned w0 = wv.x, w1 = wv.y, w2 = wv.z;
    const int g0 = (int)((w0 & 0xff) * a.cb + ((w0 >> 8) & 0xff) * a.cg + ((w0 >> 16) & 0xff) * a.cr + a.rnd) >> a.shift;
    const int g1 = (int)((w0 >> 24) * a.cb + (w1 & 0xff) * a.cg + ((w1 >> 8) & 0xff) * a.cr + a.rnd) >> a.shift;
    const int g2 = (int)(((w1 >> 16) & 0xff) * a.cb + (w1 >> 24) * a.cg + (w2 & 0xff) * a.cr + a.rnd) >> a.shift;
    const int g3 = (int)(((w2 >> 8) & 0xff) * a.cb + ((w2 >> 16) & 0xff) * a.cg + (w2 >> 24) * a.cr + a.rnd) >> a.shift;
    dst[i] = (unsigned)g0 | ((unsigned)g1 << 8) | ((unsigned)g2 << 16) | ((unsigned)g3 << 24);
  }
}

// VEC: four pixels per thread (k_gray4's arithmetic and requirements) or one (k_gray's)
template <bool VEC>
__global__ __launch_bounds__(256) void k_gray_tab(GrayArgs a, TablesArg t) {
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    const int i = threadIdx.x;
    if (i < t.nf) t.d_frames[i] = t.frames[i];
    if (i < t.npairs) {
      t.d_outs[i] = t.outs[i];
      t.d_pairs[2 * i] = t.pairs[2 * i];
      t.d_pairs[2 * i + 1] = t.pairs[2 * i + 1];
    }
  }
  const uint8_t* __restrict__ frame = st_gl(t.frames[blockIdx.y]);
  uint8_t* __restrict__ out = a.gray + (size_t)blockIdx.y * a.npix;
  if (VEC) {
    gray4_body(reinterpret_cast<const unsigned*>(frame), reinterpret_cast<unsigned*>(out), a.npix >> 2, a);
  } else {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.npix; i += gridDim.x * blockDim.x) {
      int c0 = frame[3 * (size_t)i], c1 = frame[3 * (size_t)i + 1], c2 = frame[3 * (size_t)i + 2];
      out[i] = (uint8_t)((c0 * a.cb + c1 * a.cg + c2 * a.cr + a.rnd) >> a.shift);
    }
  }
}

// (a pass that skips the luma kernel -- ST_PYR_FOLD_GRAY -- writes its tables with this one)
__global__ __launch_bounds__(64) void k_set_tables(TablesArg a) {
  const int t = threadIdx.x;
  if (t < a.nf) a.d_frames[t] = a.frames[t];
  if (t < a.npairs) {
    a.d_outs[t] = a.outs[t];
    a.d_pairs[2 * t] = a.pairs[2 * t];
    a.d_pairs[2 * t + 1] = a.pairs[2 * t + 1];
  }
}

// ---------------------------------------------------------------------------------------------
// Pyramid image of level k: float(gray) -> GaussianBlur(ks, sigma) at FULL resolution ->
// resize to (dh, dw).  Only the blurred samples the resize reads are computed: per 32x8
// output tile, phase 1 row-filters the needed sample columns of the needed source rows
// into LDS, phase 2 column-filters and interpolates.
// ---------------------------------------------------------------------------------------------
enum { PYR_COPY = 0, PYR_AREA2 = 1, PYR_LINEAR = 2 };
constexpr int PYR_OW = 32, PYR_OH = 8;

struct PyrArgs {
  const uint8_t* gray;  // n x (sh*sw)
  float* img;           // n x (dh*dw)
  int sh, sw, dh, dw;
  int ks, mode, max_rows, max_cols;
  double scale_x, scale_y;
  float taps[kMaxTaps];
};

__device__ __forceinline__ void pyr_sample(int d, int slen, double scale, int mode, int* s0, int* s1, float* f) {
  if (mode == PYR_COPY) { *s0 = *s1 = d; *f = 0.f; return; }
  if (mode == PYR_AREA2) { *s0 = 2 * d; *s1 = 2 * d + 1; *f = 0.5f; return; }
  float fx = (float)((d + 0.5) * scale - 0.5);
  int sx = (int)floorf(fx);
  fx -= sx;
  *s0 = sx; *s1 = sx + 1; *f = fx;
}

// clamped sample columns of output column d (horizontal clamp rule of cv::resize for LINEAR)
__device__ __forceinline__ void pyr_xcols(int d, const PyrArgs& a, int* c0, int* c1) {
  int xs0, xs1; float fx;
  pyr_sample(d, a.sw, a.scale_x, a.mode, &xs0, &xs1, &fx);
  if (a.mode == PYR_LINEAR) {
    if (xs0 < 0) xs0 = 0;
    if (xs0 >= a.sw - 1) xs0 = a.sw - 1;
    xs1 = min(xs0 + 1, a.sw - 1);
  }
  *c0 = xs0; *c1 = xs1;
}

__global__ __launch_bounds__(256) void k_pyr(PyrArgs a) {
  extern __shared__ __attribute__((aligned(16))) float hb[];  // [max_rows][2*PYR_OW] then the u8 source tile
  const int tid = threadIdx.x;
  const int ox0 = blockIdx.x * PYR_OW, oy0 = blockIdx.y * PYR_OH;
  const uint8_t* __restrict__ src = a.gray + (size_t)blockIdx.z * a.sh * a.sw;
  float* __restrict__ dst = a.img + (size_t)blockIdx.z * a.dh * a.dw;
  const int r = a.ks / 2;
  const int oy_last = min(oy0 + PYR_OH, a.dh) - 1;
  const int ox_last = min(ox0 + PYR_OW, a.dw) - 1;
  uint8_t* tile = reinterpret_cast<uint8_t*>(hb + (size_t)a.max_rows * 2 * PYR_OW);

  // source window of this tile: rows/cols of the blurred image that the resize touches, +-r
  int ys0, ys1; float fy;
  pyr_sample(oy0, a.sh, a.scale_y, a.mode, &ys0, &ys1, &fy);
  int ya0 = d_clamp(ys0, 0, a.sh - 1);
  pyr_sample(oy_last, a.sh, a.scale_y, a.mode, &ys0, &ys1, &fy);
  int ya1 = d_clamp(ys1, 0, a.sh - 1);
  const int ybase = ya0 - r;
  const int nrows = ya1 + r - ybase + 1;
  int xa0, xa1, tmp;
  pyr_xcols(ox0, a, &xa0, &tmp);
  pyr_xcols(ox_last, a, &tmp, &xa1);
  const int xbase = xa0 - r;
  const int ncols = xa1 + r - xbase + 1;
  const int pitch = a.max_cols;

  // phase 0: stage the u8 source window (borders already reflected) in LDS, coalesced along rows
  for (int idx = tid; idx < nrows * ncols; idx += 256) {
    const int j = idx / ncols, i = idx - j * ncols;
    tile[j * pitch + i] = src[(size_t)d_reflect101(ybase + j, a.sh) * a.sw + d_reflect101(xbase + i, a.sw)];
  }
  __syncthreads();

  // phase 1: row filter at the sample columns
  for (int idx = tid; idx < nrows * 2 * PYR_OW; idx += 256) {
    const int j = idx / (2 * PYR_OW), c = idx - j * (2 * PYR_OW);
    const int ox = ox0 + (c >> 1);
    float v = 0.f;
    if (ox < a.dw) {
      int c0, c1;
      pyr_xcols(ox, a, &c0, &c1);
      const uint8_t* S = tile + j * pitch + ((c & 1) ? c1 : c0) - xbase;  // S[k - r] = tap k
      if (a.ks == 3) {
        v = (float)S[0] * a.taps[1] + ((float)S[-1] + (float)S[1]) * a.taps[2];
      } else if (a.ks == 5) {
        v = (float)S[0] * a.taps[2] + ((float)S[-1] + (float)S[1]) * a.taps[3] + ((float)S[-2] + (float)S[2]) * a.taps[4];
      } else {
        v = a.taps[0] * (float)S[-r];
        for (int k = 1; k < a.ks; ++k) v += a.taps[k] * (float)S[k - r];
      }
    }
    hb[idx] = v;
  }
  __syncthreads();

  // phase 2: column filter at the sample rows, then interpolate
  const int tx = tid & (PYR_OW - 1), ty = tid / PYR_OW;
  const int ox = ox0 + tx, oy = oy0 + ty;
  if (ox >= a.dw || oy >= a.dh) return;
  pyr_sample(oy, a.sh, a.scale_y, a.mode, &ys0, &ys1, &fy);
  float bv[2][2];
  const int nsr = (a.mode == PYR_COPY) ? 1 : 2;
#pragma unroll
  for (int ry = 0; ry < 2; ++ry) {
    if (ry >= nsr) break;
    const int ys = d_clamp(ry ? ys1 : ys0, 0, a.sh - 1);
#pragma unroll
    for (int rx = 0; rx < 2; ++rx) {
      if (rx >= nsr) break;
      const float* col = hb + 2 * tx + rx;
      float s;
      if (a.ks == 3) {
        // tmp rows are indexed through reflect101 of the row index: LDS row of source row q is
        // the first window row whose reflected index equals q; the window was filled with
        // reflect101(ybase + j), so row (ys + t) lives at j = ys + t - ybase.
        const int j = ys - ybase;
        s = (col[(j - 1) * 2 * PYR_OW] + col[(j + 1) * 2 * PYR_OW]) * a.taps[2] + col[j * 2 * PYR_OW] * a.taps[1];
      } else {
        const int j = ys - ybase;
        s = a.taps[r] * col[j * 2 * PYR_OW];
        for (int k = 1; k <= r; ++k)
          s += a.taps[r + k] * (col[(j + k) * 2 * PYR_OW] + col[(j - k) * 2 * PYR_OW]);
      }
      bv[ry][rx] = s;
    }
  }
  float out;
  if (a.mode == PYR_COPY) {
    out = bv[0][0];
  } else if (a.mode == PYR_AREA2) {
    out = ((bv[0][0] + bv[0][1]) + (bv[1][0] + bv[1][1])) * 0.25f;
  } else {
    int xs0, xs1; float fx;
    pyr_sample(ox, a.sw, a.scale_x, a.mode, &xs0, &xs1, &fx);
    if (xs0 < 0) { fx = 0.f; xs0 = 0; }
    if (xs0 >= a.sw - 1) { fx = 0.f; xs0 = a.sw - 1; }
    const float a1 = fx, a0 = 1.f - fx;
    float r0, r1;
    if (xs0 + 1 < a.sw) {
      r0 = bv[0][0] * a0 + bv[0][1] * a1;
      r1 = bv[1][0] * a0 + bv[1][1] * a1;
    } else {
      r0 = bv[0][0] * 1.f;
      r1 = bv[1][0] * 1.f;
    }
    const float b0 = 1.f - fy, b1 = fy;
    out = r0 * b0 + r1 * b1;
  }
  dst[(size_t)oy * a.dw + ox] = out;
}

// ---------------------------------------------------------------------------------------------
// Level 0 of the pyramid (no resize, 3x3 kernel): a pure u8 -> f32 stream.  Each thread owns
// four adjacent columns (one dword of gray per row, one 16-byte store) and marches down a
// segment keeping the three row-filtered rows of the column filter in registers.
// Any width >= 2 (rows of w % 4 != 0 start at any byte: the dword loads and the 16-byte stores are then unaligned, which
// global memory allows; the thread that holds a row's last, partial group of columns goes bytewise) and h >= 2; 426 x 240,
// the legacy flow-histogram pipeline's size, takes this kernel instead of the generic tile kernel (216 -> 40 us per 257
// frames).
// ---------------------------------------------------------------------------------------------
struct Pyr0Args {
  const uint8_t* gray;  // n x (h*w)
  float* img;           // n x (h*w)
  int h, w, rows_per_seg;
  float k0, k1;         // centre / side tap
};

__device__ __forceinline__ void pyr0_hrow(const uint8_t* __restrict__ g, int w, int x0, float k0, float k1, float hb[4]) {
  if (x0 + 4 <= w) {
    typedef unsigned u32u __attribute__((aligned(1)));
    const unsigned q = *reinterpret_cast<const u32u*>(g + x0);
    const float c0 = (float)(q & 0xff), c1 = (float)((q >> 8) & 0xff), c2 = (float)((q >> 16) & 0xff), c3 = (float)(q >> 24);
    const float l = (float)g[x0 == 0 ? 1 : x0 - 1];
    const float r = (float)g[x0 + 4 >= w ? w - 2 : x0 + 4];
    hb[0] = c0 * k0 + (l + c1) * k1;
    hb[1] = c1 * k0 + (c0 + c2) * k1;
    hb[2] = c2 * k0 + (c1 + c3) * k1;
    hb[3] = c3 * k0 + (c2 + r) * k1;
  } else {
    // the row's last group holds 1-3 columns: byte loads with the reflected neighbours, the missing columns unused
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = x0 + j;
      hb[j] = c < w ? (float)g[c] * k0 + ((float)g[d_reflect101(c - 1, w)] + (float)g[d_reflect101(c + 1, w)]) * k1 : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void k_pyr0(Pyr0Args a) {
  const int h = a.h, w = a.w;
  const int x0 = 4 * (blockIdx.x * 256 + threadIdx.x);
  if (x0 >= w) return;
  const size_t np = (size_t)h * w;
  const uint8_t* __restrict__ g = a.gray + (size_t)blockIdx.z * np;
  float* __restrict__ o = a.img + (size_t)blockIdx.z * np;
  const int y0 = blockIdx.y * a.rows_per_seg;
  const int y1 = min(h, y0 + a.rows_per_seg);
  const bool full = x0 + 4 <= w;
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  float hm[4], hc[4], hp[4];
  pyr0_hrow(g + (size_t)d_reflect101(y0 - 1, h) * w, w, x0, a.k0, a.k1, hm);
  pyr0_hrow(g + (size_t)y0 * w, w, x0, a.k0, a.k1, hc);
  for (int y = y0; y < y1; ++y) {
    pyr0_hrow(g + (size_t)d_reflect101(y + 1, h) * w, w, x0, a.k0, a.k1, hp);
    f4u v;
    v.x = (hm[0] + hp[0]) * a.k1 + hc[0] * a.k0;
    v.y = (hm[1] + hp[1]) * a.k1 + hc[1] * a.k0;
    v.z = (hm[2] + hp[2]) * a.k1 + hc[2] * a.k0;
    v.w = (hm[3] + hp[3]) * a.k1 + hc[3] * a.k0;
    float* dst = o + (size_t)y * w + x0;
    if (full) {
      *reinterpret_cast<f4u*>(dst) = v;
    } else {
      dst[0] = v.x;
      if (x0 + 1 < w) dst[1] = v.y;
      if (x0 + 2 < w) dst[2] = v.z;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { hm[i] = hc[i]; hc[i] = hp[i]; }
  }
}

// ---------------------------------------------------------------------------------------------
// Pyramid levels with an exact power-of-two decimation S (sw == S*dw, sh == S*dh): the two
// blurred samples per axis that cv::resize combines are at S*d + OFF and S*d + OFF + 1
// (OFF = 0 for S == 2, where OpenCV takes the INTER_AREA 2x2 mean; OFF = S/2 - 1 with weights
// 1/2, 1/2 for S >= 4).  Each thread owns one output column and marches down the output rows
// keeping the KS+1 row-filtered source rows of the column filter in registers; every new source
// row costs KS+1 byte loads (served by L1, neighbours overlap).  No LDS, no barriers.
// ---------------------------------------------------------------------------------------------
struct PyrDecArgs {
  const uint8_t* gray;  // n x (sh*sw)
  float* img;           // n x (dh*dw)
  int sh, sw, dh, dw, rows_per_seg;
  float taps[kMaxTaps];
};

template <int KS>
__device__ __forceinline__ float pyr_rowfilter(const float* __restrict__ b, const float* __restrict__ k) {
  // b[0..KS-1]: source values left to right
  constexpr int r = KS / 2;
  if (KS == 3) return b[1] * k[1] + (b[0] + b[2]) * k[2];
  if (KS == 5) return b[2] * k[2] + (b[1] + b[3]) * k[3] + (b[0] + b[4]) * k[4];
  float v = k[0] * b[0];
#pragma unroll
  for (int i = 1; i < KS; ++i) v += k[i] * b[i];
  (void)r;
  return v;
}

template <int KS>
__device__ __forceinline__ float pyr_colfilter(const float* __restrict__ c, const float* __restrict__ k) {
  // c[0..KS-1]: row-filtered values top to bottom
  constexpr int r = KS / 2;
  if (KS == 3) return (c[0] + c[2]) * k[2] + c[1] * k[1];
  float s = k[r] * c[r];
#pragma unroll
  for (int i = 1; i <= r; ++i) s += k[r + i] * (c[r + i] + c[r - i]);
  return s;
}

template <int S, int KS>
__global__ __launch_bounds__(256) void k_pyr_dec(PyrDecArgs a) {
  constexpr int r = KS / 2;
  constexpr int OFF = S == 2 ? 0 : S / 2 - 1;
  static_assert(KS + 1 > S, "window must overlap between output rows");
  const int dx = blockIdx.x * 256 + threadIdx.x;
  if (dx >= a.dw) return;
  const int sw = a.sw, sh = a.sh;
  const uint8_t* __restrict__ src = a.gray + (size_t)blockIdx.z * sh * sw;
  float* __restrict__ dst = a.img + (size_t)blockIdx.z * a.dh * a.dw;
  const int xA = S * dx + OFF;
  const int xl = xA - r;  // leftmost source column of the two row filters
  const bool edge = xl < 0 || xl + KS >= sw;
  const int dy0 = blockIdx.y * a.rows_per_seg;
  const int dy1 = min(a.dh, dy0 + a.rows_per_seg);

  // Source bytes [xl, xl+KS] of a row.  Interior threads fetch them as NW aligned dwords (one or
  // two wide loads instead of KS+1 byte loads: the byte version was bound by the number of
  // vector-memory instructions) and realign with v_alignbyte; border threads, and rows that are not
  // dword aligned, take the reflected byte path.
  constexpr int NW = S == 2 ? 2 : (S == 4 ? 3 : 6);
  constexpr int NAL = (KS + 1 + 3) / 4;
  const int xa = xl & ~3;
  const unsigned shift = (unsigned)(xl & 3);
  const bool wide = !edge && (sw & 3) == 0 && xa + 4 * NW <= sw && (((uintptr_t)src) & 3) == 0;
  auto hrow = [&](int y, float& hA, float& hB) {
    const uint8_t* __restrict__ g = src + (size_t)d_reflect101(y, sh) * sw;
    float b[KS + 1];
    if (wide) {
      const unsigned* __restrict__ gw = reinterpret_cast<const unsigned*>(g + xa);
      unsigned wv[NW];
#pragma unroll
      for (int i = 0; i < NW; ++i) wv[i] = gw[i];
      unsigned al[NAL];
#pragma unroll
      for (int j = 0; j < NAL; ++j) al[j] = __builtin_amdgcn_alignbyte(wv[j + 1 < NW ? j + 1 : NW - 1], wv[j], shift);
#pragma unroll
      for (int i = 0; i <= KS; ++i) b[i] = (float)((al[i >> 2] >> (8 * (i & 3))) & 0xffu);
    } else if (!edge) {
#pragma unroll
      for (int i = 0; i <= KS; ++i) b[i] = (float)g[xl + i];
    } else {
#pragma unroll
      for (int i = 0; i <= KS; ++i) b[i] = (float)g[d_reflect101(xl + i, sw)];
    }
    hA = pyr_rowfilter<KS>(b, a.taps);
    hB = pyr_rowfilter<KS>(b + 1, a.taps);
  };

  // ring[j] = row-filtered source row (S*dy + OFF - r + j), j = 0..KS.  One source row enters
  // per loop trip (the ring shifts by one register); an output row is emitted every S trips once
  // the ring is full.  The loop is deliberately not unrolled: it bounds the loads in flight.
  float hA[KS + 1], hB[KS + 1];
#pragma unroll
  for (int j = 0; j <= KS; ++j) hA[j] = hB[j] = 0.f;
  const int ys0 = S * dy0 + OFF - r;
  const int nsrc = KS + 1 + S * (dy1 - dy0 - 1);
  int dy = dy0, until = KS + 1;
#pragma unroll 1
  for (int t = 0; t < nsrc; ++t) {
    float nA, nB;
    hrow(ys0 + t, nA, nB);
#pragma unroll
    for (int j = 0; j < KS; ++j) { hA[j] = hA[j + 1]; hB[j] = hB[j + 1]; }
    hA[KS] = nA; hB[KS] = nB;
    if (t + 1 == until) {
      until += S;
      const float b00 = pyr_colfilter<KS>(hA, a.taps), b01 = pyr_colfilter<KS>(hB, a.taps);
      const float b10 = pyr_colfilter<KS>(hA + 1, a.taps), b11 = pyr_colfilter<KS>(hB + 1, a.taps);
      float out;
      if (S == 2) {
        out = ((b00 + b01) + (b10 + b11)) * 0.25f;
      } else {
        const float r0 = b00 * 0.5f + b01 * 0.5f, r1 = b10 * 0.5f + b11 * 0.5f;
        out = r0 * 0.5f + r1 * 0.5f;
      }
      dst[(size_t)dy * a.dw + dx] = out;
      ++dy;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// All four pyramid levels of the reference's default geometry in ONE pass over the gray frame
// (levels 0..3 = blur 3/3/9/19 taps + decimation by 1/2/4/8; frame sides multiples of 8).  The
// per-level kernels above read the gray frame four times and, marching one source row per trip
// with one or two loads in flight, are bound by memory latency; here a workgroup owns a strip of
// 1024 source columns, stages 8 source rows per barrier through LDS (one aligned dword per thread
// and row, the next batch's loads in flight while the current one is consumed) and every thread
// feeds the row-filter rings of all four levels from the staged bytes:
//   level 0: thread t -> columns 4t..4t+3          level 1: outputs 2t, 2t+1
//   level 2: output t                               level 3: output u = 32*wave + lane%32, the two
//                                                   blurred columns split over lane halves
// Arithmetic (operand order, association) is that of k_pyr0 / k_pyr_dec, so the results are
// bit-identical to theirs.
// ---------------------------------------------------------------------------------------------
struct PyrFusedArgs {
  const uint8_t* gray;          // n x (h*w), or
  const uint8_t* const* frames; // RGB instance: device table of n (h,w,3) frames, converted on the fly
  int cb, cg, cr, rnd, shift;   // luma table of the RGB instance (see launch_gray)
  float* img0; float* img1; float* img2; float* img3;  // n x (h>>k)*(w>>k)
  int h, w, rows_per_seg;       // rows_per_seg: source rows per segment, multiple of 8
  int strip_w;                  // source columns per workgroup strip
  // taps of levels 0..3 (3, 3, 9, 19), first half only: cv::getGaussianKernel is exactly symmetric
  // (k[i] == k[n-1-i]), and 19 scalars instead of 34 stay within the scalar register file
  float k0[2], k1[2], k2[5], k3[10];
};

// pyr_rowfilter / pyr_colfilter with the taps given by their first half (kh[i] = k[i], i <= KS/2)
template <int KS>
__device__ __forceinline__ float pf_rowfilter(const float* __restrict__ b, const float* __restrict__ kh) {
  constexpr int r = KS / 2;
  if (KS == 3) return b[1] * kh[1] + (b[0] + b[2]) * kh[0];
  float v = kh[0] * b[0];
#pragma unroll
  for (int i = 1; i < KS; ++i) v += kh[i <= r ? i : KS - 1 - i] * b[i];
  return v;
}
template <int KS>
__device__ __forceinline__ float pf_colfilter(const float* __restrict__ c, const float* __restrict__ kh) {
  constexpr int r = KS / 2;
  if (KS == 3) return (c[0] + c[2]) * kh[0] + c[1] * kh[1];
  float s = kh[r] * c[r];
#pragma unroll
  for (int i = 1; i <= r; ++i) s += kh[r - i] * (c[r + i] + c[r - i]);
  return s;
}

constexpr int PF_RB = 8;          // source rows per barrier
constexpr int PF_ROWDW = 260;     // staged row: 8 halo bytes + 1024 + 8 halo bytes, as dwords

// four source bytes at columns col0..col0+3 with BORDER_REFLECT_101 (frame borders only: rare)
__device__ __forceinline__ unsigned pf_load4_border(const uint8_t* __restrict__ row, int col0, int w) {
  unsigned v = 0;
#pragma unroll 1
  for (int k = 0; k < 4; ++k) v |= (unsigned)row[d_reflect101(col0 + k, w)] << (8 * k);
  return v;
}
__device__ __forceinline__ unsigned pf_load4(const uint8_t* __restrict__ row, int col0, int w) {
  if (col0 >= 0 && col0 + 3 < w) return *reinterpret_cast<const unsigned*>(row + col0);
  if (col0 >= w + 8 || col0 < -8) return 0u;  // never read
  return pf_load4_border(row, col0, w);
}

__device__ __forceinline__ float pf_byte(unsigned d, int k) { return (float)((d >> (8 * k)) & 0xffu); }

// The same four gray bytes computed from the RGB frame (k_gray4's arithmetic: OpenCV's BGR table on RGB
// bytes), so that the gray image never goes through memory: 12 source bytes as three aligned dwords.
// Needs 4-byte aligned frames and w % 4 == 0 (checked on the host).
__device__ __forceinline__ unsigned pf_gray1(const uint8_t* __restrict__ px, const PyrFusedArgs& a) {
  return (unsigned)((px[0] * a.cb + px[1] * a.cg + px[2] * a.cr + a.rnd) >> a.shift);
}
__device__ __forceinline__ unsigned pf_load4_rgb(const uint8_t* __restrict__ row, int col0, int w, const PyrFusedArgs& a) {
  if (col0 >= 0 && col0 + 3 < w) {
    const unsigned* __restrict__ q = reinterpret_cast<const unsigned*>(row + 3 * col0);
    const unsigned w0 = q[0], w1 = q[1], w2 = q[2];
    const unsigned g0 = ((w0 & 0xff) * a.cb + ((w0 >> 8) & 0xff) * a.cg + ((w0 >> 16) & 0xff) * a.cr + a.rnd) >> a.shift;
    const unsigned g1 = ((w0 >> 24) * a.cb + (w1 & 0xff) * a.cg + ((w1 >> 8) & 0xff) * a.cr + a.rnd) >> a.shift;
    const unsigned g2 = (((w1 >> 16) & 0xff) * a.cb + (w1 >> 24) * a.cg + (w2 & 0xff) * a.cr + a.rnd) >> a.shift;
    const unsigned g3 = (((w2 >> 8) & 0xff) * a.cb + ((w2 >> 16) & 0xff) * a.cg + (w2 >> 24) * a.cr + a.rnd) >> a.shift;
    return g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
  }
  if (col0 >= w + 8 || col0 < -8) return 0u;  // never read
  unsigned v = 0;
#pragma unroll 1
  for (int k = 0; k < 4; ++k) v |= pf_gray1(row + 3 * d_reflect101(col0 + k, w), a) << (8 * k);
  return v;
}

template <bool RGB>
__global__ __launch_bounds__(256) void k_pyr_fused(PyrFusedArgs a) {
  __shared__ unsigned srow[2][PF_RB][PF_ROWDW];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int h = a.h, w = a.w;
  const int SW = a.strip_w;           // source columns per strip: multiple of 8, <= 1024
  const int X0 = blockIdx.x * SW;
  const int Y0 = blockIdx.y * a.rows_per_seg, Y1 = min(h, Y0 + a.rows_per_seg);
  const size_t np = (size_t)h * w;
  const uint8_t* __restrict__ g = RGB ? st_gl(a.frames[blockIdx.z]) : a.gray + (size_t)blockIdx.z * np;
  const int gw = RGB ? 3 * w : w;  // bytes per source row
  float* __restrict__ o0 = a.img0 + (size_t)blockIdx.z * np;
  float* __restrict__ o1 = a.img1 + (size_t)blockIdx.z * (np >> 2);
  float* __restrict__ o2 = a.img2 + (size_t)blockIdx.z * (np >> 4);
  float* __restrict__ o3 = a.img3 + (size_t)blockIdx.z * (np >> 6);
  const int u = wv * 32 + (lane & 31), half = lane >> 5;  // level-3 output and blurred column
  const bool live = 4 * t < SW && X0 + 4 * t < w;          // this thread's columns exist
  const bool live3 = 8 * u < SW && X0 + 8 * u < w;
  // the 4 halo dwords of each of the 8 staged rows are fetched by threads 0..31 (row t/4, dword t%4)
  const int hq = t & 3, hrow = (t >> 2) & 7;
  const int hcol = hq < 2 ? X0 - 8 + 4 * hq : X0 + SW + 4 * (hq - 2);
  const int hidx = hq < 2 ? hq : 2 + (SW >> 2) + (hq - 2);

  float hm[4], hc[4];                 // level 0: row-filtered rows y-2, y-1
  float r1A[2][4], r1B[2][4];         // level 1: KS+1 = 4 row-filtered rows per output and column
  float r2A[10], r2B[10];             // level 2: KS+1 = 10
  float r3[20];                       // level 3: KS+1 = 20 (one blurred column per lane half)
#pragma unroll
  for (int j = 0; j < 4; ++j) { hm[j] = hc[j] = 0.f; r1A[0][j] = r1A[1][j] = r1B[0][j] = r1B[1][j] = 0.f; }
#pragma unroll
  for (int j = 0; j < 10; ++j) r2A[j] = r2B[j] = 0.f;
#pragma unroll
  for (int j = 0; j < 20; ++j) r3[j] = 0.f;

  unsigned pre[PF_RB], preh = 0u;
  auto fetch = [&](int ybase) {
#pragma unroll
    for (int r = 0; r < PF_RB; ++r)
      pre[r] = RGB ? pf_load4_rgb(g + (size_t)d_reflect101(ybase + r, h) * gw, X0 + 4 * t, w, a)
                   : pf_load4(g + (size_t)d_reflect101(ybase + r, h) * gw, X0 + 4 * t, w);
    if (t < 32) preh = RGB ? pf_load4_rgb(g + (size_t)d_reflect101(ybase + hrow, h) * gw, hcol, w, a)
                           : pf_load4(g + (size_t)d_reflect101(ybase + hrow, h) * gw, hcol, w);
  };
  auto stash = [&](int buf) {
    if (4 * t < SW) {  // threads beyond the strip would overwrite its right halo
#pragma unroll
      for (int r = 0; r < PF_RB; ++r) srow[buf][r][2 + t] = pre[r];
    }
    if (t < 32) srow[buf][hrow][hidx] = preh;
  };

  // source rows Y0-8 .. Y1+7 feed the outputs of [Y0, Y1) at every level (level 3 reaches 6 rows
  // above and 5 below its block of 8)
  const int nb = (Y1 - Y0) / PF_RB + 2;
  fetch(Y0 - PF_RB);
  stash(0);
  __syncthreads();
  for (int b = 0; b < nb; ++b) {
    const int ybase = Y0 - PF_RB + b * PF_RB;
    const int buf = b & 1;
    if (b + 1 < nb) fetch(ybase + PF_RB);
    // fully unrolled: with r a compile-time constant the emission tests below fold away (a rolled
    // loop measured 20 % slower although its body fits the instruction cache more easily)
#pragma unroll
    for (int r = 0; r < PF_RB; ++r) {
      const int y = ybase + r;  // source row entering (unreflected index)
      const unsigned* __restrict__ s = srow[buf][r];
      // bytes of columns X0+4t-4 .. X0+4t+7
      const unsigned d0 = s[t + 1], d1 = s[t + 2], d2 = s[t + 3];
      float bb[12];
#pragma unroll
      for (int k = 0; k < 4; ++k) { bb[k] = pf_byte(d0, k); bb[4 + k] = pf_byte(d1, k); bb[8 + k] = pf_byte(d2, k); }
      // ---- level 0 (k_pyr0)
      {
        float hp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) hp[j] = bb[4 + j] * a.k0[1] + (bb[3 + j] + bb[5 + j]) * a.k0[0];
        const int yo = y - 1;
        if (live && yo >= Y0 && yo < Y1) {
          float4 v;
          v.x = (hm[0] + hp[0]) * a.k0[0] + hc[0] * a.k0[1];
          v.y = (hm[1] + hp[1]) * a.k0[0] + hc[1] * a.k0[1];
          v.z = (hm[2] + hp[2]) * a.k0[0] + hc[2] * a.k0[1];
          v.w = (hm[3] + hp[3]) * a.k0[0] + hc[3] * a.k0[1];
          *reinterpret_cast<float4*>(o0 + (size_t)yo * w + X0 + 4 * t) = v;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { hm[j] = hc[j]; hc[j] = hp[j]; }
      }
      // ---- level 1 (k_pyr_dec<2,3>): outputs 2t+e, blurred columns 4t+2e and 4t+2e+1
      {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float nA = pf_rowfilter<3>(bb + 3 + 2 * e, a.k1), nB = pf_rowfilter<3>(bb + 4 + 2 * e, a.k1);
#pragma unroll
          for (int j = 0; j < 3; ++j) { r1A[e][j] = r1A[e][j + 1]; r1B[e][j] = r1B[e][j + 1]; }
          r1A[e][3] = nA; r1B[e][3] = nB;
        }
        if ((r & 1) == 0) {  // y = 2*dy + 2
          const int dy = (y - 2) >> 1;
          if (live && dy >= (Y0 >> 1) && dy < (Y1 >> 1)) {
            float2 v;
            {
              const float b00 = pf_colfilter<3>(r1A[0], a.k1), b01 = pf_colfilter<3>(r1B[0], a.k1);
              const float b10 = pf_colfilter<3>(r1A[0] + 1, a.k1), b11 = pf_colfilter<3>(r1B[0] + 1, a.k1);
              v.x = ((b00 + b01) + (b10 + b11)) * 0.25f;
            }
            {
              const float b00 = pf_colfilter<3>(r1A[1], a.k1), b01 = pf_colfilter<3>(r1B[1], a.k1);
              const float b10 = pf_colfilter<3>(r1A[1] + 1, a.k1), b11 = pf_colfilter<3>(r1B[1] + 1, a.k1);
              v.y = ((b00 + b01) + (b10 + b11)) * 0.25f;
            }
            *reinterpret_cast<float2*>(o1 + (size_t)dy * (w >> 1) + (X0 >> 1) + 2 * t) = v;
          }
        }
      }
      // ---- level 2 (k_pyr_dec<4,9>): output t, blurred columns 4t+1 and 4t+2, window from 4t-3
      {
        const float nA = pf_rowfilter<9>(bb + 1, a.k2), nB = pf_rowfilter<9>(bb + 2, a.k2);
#pragma unroll
        for (int j = 0; j < 9; ++j) { r2A[j] = r2A[j + 1]; r2B[j] = r2B[j + 1]; }
        r2A[9] = nA; r2B[9] = nB;
        if ((r & 3) == 2) {  // y = 4*dy + 6
          const int dy = (y - 6) >> 2;
          if (live && dy >= (Y0 >> 2) && dy < (Y1 >> 2)) {
            const float b00 = pf_colfilter<9>(r2A, a.k2), b01 = pf_colfilter<9>(r2B, a.k2);
            const float b10 = pf_colfilter<9>(r2A + 1, a.k2), b11 = pf_colfilter<9>(r2B + 1, a.k2);
            const float q0 = b00 * 0.5f + b01 * 0.5f, q1 = b10 * 0.5f + b11 * 0.5f;
            o2[(size_t)dy * (w >> 2) + (X0 >> 2) + t] = q0 * 0.5f + q1 * 0.5f;
          }
        }
      }
      // ---- level 3 (k_pyr_dec<8,19>): output u, blurred column 8u+3+half, window from 8u-6+half
      {
        // dwords covering columns X0+8u-8 .. X0+8u+15, realigned by `half` bytes so that both lane
        // halves run the same code: cb[k] = column X0+8u-8+half+k
        unsigned dw[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) dw[q] = s[2 * u + q];
        float cb[21];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const unsigned d = __builtin_amdgcn_alignbyte(dw[q + 1 < 6 ? q + 1 : 5], dw[q], (unsigned)half);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (4 * q + k < 21) cb[4 * q + k] = pf_byte(d, k);
        }
        const float n3 = pf_rowfilter<19>(cb + 2, a.k3);
#pragma unroll
        for (int j = 0; j < 19; ++j) r3[j] = r3[j + 1];
        r3[19] = n3;
        if (r == 5) {  // y = 8*dy + 13
          const int dy = (y - 13) >> 3;
          if (dy >= (Y0 >> 3) && dy < (Y1 >> 3)) {  // uniform over the workgroup
            const float c0 = pf_colfilter<19>(r3, a.k3), c1 = pf_colfilter<19>(r3 + 1, a.k3);
            const float p0 = __shfl_xor(c0, 32), p1 = __shfl_xor(c1, 32);  // the other blurred column
            if (half == 0 && live3) {
              const float q0 = c0 * 0.5f + p0 * 0.5f, q1 = c1 * 0.5f + p1 * 0.5f;
              o3[(size_t)dy * (w >> 3) + (X0 >> 3) + u] = q0 * 0.5f + q1 * 0.5f;
            }
          }
        }
      }
    }
    if (b + 1 < nb) stash(buf ^ 1);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// The same one-pass pyramid with the levels split over three ROLES of four waves each (768-thread workgroups):
// role 0 = levels 0 and 1, role 1 = level 2, role 2 = level 3, all reading the same staged rows.  A gfx950 wave issues
// at most one vector-ALU instruction per ~8 clocks while a SIMD can issue one per 4 (scripts/ubench/valu1wave.hip), so a
// launch that leaves one wave per SIMD -- a call of one or a few pairs -- runs at half the ALU rate; here the
// instruction stream of a strip segment is cut in three (~120 / 90 / 110 instructions per source row instead of ~330
// in one wave), three waves share a SIMD, and each role keeps only its own rings in registers.  Arithmetic and
// association are those of k_pyr_fused (and of k_pyr0 / k_pyr_dec): bit-identical images.
// ---------------------------------------------------------------------------------------------
template <int ROLE, bool L0 = true>
__device__ __forceinline__ void pyr_role_run(const PyrFusedArgs& a, unsigned (*srow)[PF_RB][PF_ROWDW]) {
  const int t = threadIdx.x & 255, lane = t & 63, wv = t >> 6;
  const int h = a.h, w = a.w;
  const int SW = a.strip_w;
  const int X0 = blockIdx.x * SW;
  const int Y0 = blockIdx.y * a.rows_per_seg, Y1 = min(h, Y0 + a.rows_per_seg);
  const size_t np = (size_t)h * w;
  const uint8_t* __restrict__ g = a.gray + (size_t)blockIdx.z * np;
  float* __restrict__ o0 = a.img0 + (size_t)blockIdx.z * np;
  float* __restrict__ o1 = a.img1 + (size_t)blockIdx.z * (np >> 2);
  float* __restrict__ o2 = a.img2 + (size_t)blockIdx.z * (np >> 4);
  float* __restrict__ o3 = a.img3 + (size_t)blockIdx.z * (np >> 6);
  const int u = wv * 32 + (lane & 31), half = lane >> 5;
  const bool live = 4 * t < SW && X0 + 4 * t < w;
  const bool live3 = 8 * u < SW && X0 + 8 * u < w;
  const int hq = t & 3, hrow = (t >> 2) & 7;
  const int hcol = hq < 2 ? X0 - 8 + 4 * hq : X0 + SW + 4 * (hq - 2);
  const int hidx = hq < 2 ? hq : 2 + (SW >> 2) + (hq - 2);

  // staging: role r fetches rows r, r + 3, r + 6 (< 8) of a batch, dword column t; role 0's threads 0..31 the halo
  unsigned pre[3], preh = 0u;
  auto fetch = [&](int ybase) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int r = ROLE + 3 * j;
      if (r < PF_RB) pre[j] = pf_load4(g + (size_t)d_reflect101(ybase + r, h) * w, X0 + 4 * t, w);
    }
    if (ROLE == 0 && t < 32) preh = pf_load4(g + (size_t)d_reflect101(ybase + hrow, h) * w, hcol, w);
  };
  auto stash = [&](int buf) {
    if (4 * t < SW) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int r = ROLE + 3 * j;
        if (r < PF_RB) srow[buf][r][2 + t] = pre[j];
      }
    }
    if (ROLE == 0 && t < 32) srow[buf][hrow][hidx] = preh;
  };

  float hm[4], hc[4], r1A[2][4], r1B[2][4];  // role 0
  float r2A[10], r2B[10];                    // role 1
  float r3[20];                              // role 2
  if (ROLE == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { hm[j] = hc[j] = 0.f; r1A[0][j] = r1A[1][j] = r1B[0][j] = r1B[1][j] = 0.f; }
  } else if (ROLE == 1) {
#pragma unroll
    for (int j = 0; j < 10; ++j) r2A[j] = r2B[j] = 0.f;
  } else {
#pragma unroll
    for (int j = 0; j < 20; ++j) r3[j] = 0.f;
  }

  const int nb = (Y1 - Y0) / PF_RB + 2;
  fetch(Y0 - PF_RB);
  stash(0);
  __syncthreads();
  for (int b = 0; b < nb; ++b) {
    const int ybase = Y0 - PF_RB + b * PF_RB;
    const int buf = b & 1;
    if (b + 1 < nb) fetch(ybase + PF_RB);
#pragma unroll
    for (int r = 0; r < PF_RB; ++r) {
      const int y = ybase + r;
      const unsigned* __restrict__ s = srow[buf][r];
      if (ROLE == 0) {
        const unsigned d0 = s[t + 1], d1 = s[t + 2], d2 = s[t + 3];
        float bb[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) { bb[k] = pf_byte(d0, k); bb[4 + k] = pf_byte(d1, k); bb[8 + k] = pf_byte(d2, k); }
        if (L0) {  // level 0 (L0 = false: k_polyexp takes it from the gray frame itself, see polyexp_body<.., U8>)
          float hp[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) hp[j] = bb[4 + j] * a.k0[1] + (bb[3 + j] + bb[5 + j]) * a.k0[0];
          const int yo = y - 1;
          if (live && yo >= Y0 && yo < Y1) {
            float4 v;
            v.x = (hm[0] + hp[0]) * a.k0[0] + hc[0] * a.k0[1];
            v.y = (hm[1] + hp[1]) * a.k0[0] + hc[1] * a.k0[1];
            v.z = (hm[2] + hp[2]) * a.k0[0] + hc[2] * a.k0[1];
            v.w = (hm[3] + hp[3]) * a.k0[0] + hc[3] * a.k0[1];
            *reinterpret_cast<float4*>(o0 + (size_t)yo * w + X0 + 4 * t) = v;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) { hm[j] = hc[j]; hc[j] = hp[j]; }
        }
        {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float nA = pf_rowfilter<3>(bb + 3 + 2 * e, a.k1), nB = pf_rowfilter<3>(bb + 4 + 2 * e, a.k1);
#pragma unroll
            for (int j = 0; j < 3; ++j) { r1A[e][j] = r1A[e][j + 1]; r1B[e][j] = r1B[e][j + 1]; }
            r1A[e][3] = nA; r1B[e][3] = nB;
          }
          if ((r & 1) == 0) {
            const int dy = (y - 2) >> 1;
            if (live && dy >= (Y0 >> 1) && dy < (Y1 >> 1)) {
              float2 v;
              {
                const float b00 = pf_colfilter<3>(r1A[0], a.k1), b01 = pf_colfilter<3>(r1B[0], a.k1);
                const float b10 = pf_colfilter<3>(r1A[0] + 1, a.k1), b11 = pf_colfilter<3>(r1B[0] + 1, a.k1);
                v.x = ((b00 + b01) + (b10 + b11)) * 0.25f;
              }
              {
                const float b00 = pf_colfilter<3>(r1A[1], a.k1), b01 = pf_colfilter<3>(r1B[1], a.k1);
                const float b10 = pf_colfilter<3>(r1A[1] + 1, a.k1), b11 = pf_colfilter<3>(r1B[1] + 1, a.k1);
                v.y = ((b00 + b01) + (b10 + b11)) * 0.25f;
              }
              *reinterpret_cast<float2*>(o1 + (size_t)dy * (w >> 1) + (X0 >> 1) + 2 * t) = v;
            }
          }
        }
      } else if (ROLE == 1) {
        const unsigned d0 = s[t + 1], d1 = s[t + 2], d2 = s[t + 3];
        float bb[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) { bb[k] = pf_byte(d0, k); bb[4 + k] = pf_byte(d1, k); bb[8 + k] = pf_byte(d2, k); }
        const float nA = pf_rowfilter<9>(bb + 1, a.k2), nB = pf_rowfilter<9>(bb + 2, a.k2);
#pragma unroll
        for (int j = 0; j < 9; ++j) { r2A[j] = r2A[j + 1]; r2B[j] = r2B[j + 1]; }
        r2A[9] = nA; r2B[9] = nB;
        if ((r & 3) == 2) {
          const int dy = (y - 6) >> 2;
          if (live && dy >= (Y0 >> 2) && dy < (Y1 >> 2)) {
            const float b00 = pf_colfilter<9>(r2A, a.k2), b01 = pf_colfilter<9>(r2B, a.k2);
            const float b10 = pf_colfilter<9>(r2A + 1, a.k2), b11 = pf_colfilter<9>(r2B + 1, a.k2);
            const float q0 = b00 * 0.5f + b01 * 0.5f, q1 = b10 * 0.5f + b11 * 0.5f;
            o2[(size_t)dy * (w >> 2) + (X0 >> 2) + t] = q0 * 0.5f + q1 * 0.5f;
          }
        }
      } else {
        unsigned dw[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) dw[q] = s[2 * u + q];
        float cb[21];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const unsigned d = __builtin_amdgcn_alignbyte(dw[q + 1 < 6 ? q + 1 : 5], dw[q], (unsigned)half);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (4 * q + k < 21) cb[4 * q + k] = pf_byte(d, k);
        }
        const float n3 = pf_rowfilter<19>(cb + 2, a.k3);
#pragma unroll
        for (int j = 0; j < 19; ++j) r3[j] = r3[j + 1];
        r3[19] = n3;
        if (r == 5) {
          const int dy = (y - 13) >> 3;
          if (dy >= (Y0 >> 3) && dy < (Y1 >> 3)) {
            const float c0 = pf_colfilter<19>(r3, a.k3), c1 = pf_colfilter<19>(r3 + 1, a.k3);
            const float p0 = __shfl_xor(c0, 32), p1 = __shfl_xor(c1, 32);
            if (half == 0 && live3) {
              const float q0 = c0 * 0.5f + p0 * 0.5f, q1 = c1 * 0.5f + p1 * 0.5f;
              o3[(size_t)dy * (w >> 3) + (X0 >> 3) + u] = q0 * 0.5f + q1 * 0.5f;
            }
          }
        }
      }
    }
    if (b + 1 < nb) stash(buf ^ 1);
    __syncthreads();
  }
}

template <bool L0>
__global__ __launch_bounds__(768) void k_pyr_roles(PyrFusedArgs a) {
  __shared__ unsigned srow[2][PF_RB][PF_ROWDW];
  const int role = threadIdx.x >> 8;  // wave-uniform; every role passes the same barriers
  if (role == 0) pyr_role_run<0, L0>(a, srow);
  else if (role == 1) pyr_role_run<1>(a, srow);
  else pyr_role_run<2>(a, srow);
}

typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
// Loads address an expansion as (uniform base pointer) + (unsigned 32-bit BYTE offset) so that the
// compiler can use the SGPR-base addressing form (one VGPR per address, no 64-bit VALU math).
// Buffer addressing: (resource over a uniform base) + (per-thread byte offset, VGPR) + (uniform byte offset, SGPR).  A
// thread that walks down one column keeps ONE constant offset register and the row enters through the scalar operand:
// no vector arithmetic per access (the global_load form needs base + offset assembled per access once the row term is
// not foldable).  Raw buffers: stride 0, range-checked against `bytes` (out-of-range reads return 0, writes are dropped).
typedef float f4v16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t st_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes > 0xfffffffcull ? 0xfffffffcull : bytes), 0x00020000);
}
__device__ __forceinline__ float bld1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
// AUX: cache policy bits of the instruction (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX = 0>
__device__ __forceinline__ void bst1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, AUX);
}
// 16-byte store: NO scalar offset operand.  Measured on gfx950 (round 5, scripts/diag_polyexp2.py): buffer_store_dwordx4 with
// an SGPR soffset, followed one scalar instruction later by a vector write to the first two data registers, stores the NEW
// contents of those registers when the memory pipe is backed up (several workgroups per compute unit) -- the compiler's
// hazard recogniser inserts wait states for >64-bit store data only when soffset is NOT a register (the documented rule).
// With the row folded into the vector offset the store is the case the compiler protects.
template <int AUX = 0>
__device__ __forceinline__ void bst4(__amdgpu_buffer_rsrc_t r, unsigned voff, float4 v) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u4 q;
  q.x = __builtin_bit_cast(unsigned, v.x); q.y = __builtin_bit_cast(unsigned, v.y);
  q.z = __builtin_bit_cast(unsigned, v.z); q.w = __builtin_bit_cast(unsigned, v.w);
  __builtin_amdgcn_raw_buffer_store_b128(q, r, (int)voff, 0, AUX);
}

__device__ __forceinline__ float ldf(const float* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ f2u ldf2(const float* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const f2u*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float4 ldf4(const float* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}

// build-time experiment switches (0 in the product; ST_ABLATE is described above k_flow_iter3)
#ifndef ST_ABLATE
#define ST_ABLATE 0
#endif



// ---------------------------------------------------------------------------------------------
// A4: polynomial expansion.  Each thread owns one column of a 256-wide strip and marches down
// a vertical segment keeping the 2N+1 source rows of its column in registers; the three
// vertically filtered values go through LDS for the horizontal pass.
// ---------------------------------------------------------------------------------------------
struct PolyArgs {
  const float* img;  // n x (h*w)
  const uint8_t* gray;  // k_polyexp_u8: n x (h*w) gray bytes; level 0's 3x3 blur is evaluated on the fly
  float* R;          // n x (h*w float4 + h*w float), see "R layout" above update_matrices_px
  int h, w, rows_per_seg;
  PolyCoef c;
};

constexpr int PE_RB = 4;     // rows per barrier
// Output columns per 256-thread strip (<= 256 - 2*7).  240 rather than the 246 the halo allows:
// strip starts are then 128-byte aligned in the float4 plane (3840 B per strip row) and 1920, 960,
// 480, 240 are whole numbers of strips; measured 8 % faster (the stores are what the DP-heavy
// horizontal pass fails to overlap: 3.6 ms without stores, 4.4 ms with, per 257 1080p frames).
constexpr int PE_OUT = 240;

// One (strip bx, segment by) of frame z of one level; A carries the coefficients (a.c)
//
// Addressing and control flow (round 5; a tenth of the kernel's vector instructions were addresses, not arithmetic, and
// every batch ended waiting for its own stores):
// (i) LDS layout [pixel][row of the batch][sum] with a pixel stride of 13 words (odd: no bank conflicts): the 33 values
//     the horizontal pass reads for a row lie within 142 words of ONE per-thread base, inside the 8-bit immediate offsets
//     of ds_read2_b32 (before: three planes 1 KB apart -- 12 address additions per pixel); 26 KB per workgroup, six
//     workgroups per compute unit as before.
// (ii) Source and output rows through buffer instructions: the thread's column is a constant vector offset, the row the
//     scalar offset (the 16-byte store takes the row in its vector offset: see bst4).
// (iii) No branch around the horizontal pass and its stores: a lane or row that must not write is given an offset / a
//     resource the range check rejects.  With the same eight stores behind the four prefetch loads on every path the
//     compiler's wait for the prefetched rows counts them exactly (vmcnt(8..11)); with a branch it has to assume none
//     was issued and waits for the batch's own stores (vmcnt retires in order).
// Same values, same operations in the same order: bit-identical results.  Measured per 257 frames of 1080p, same box:
// 4.38 -> 4.25 (i, ii) -> 4.21 ms (iii).
// U8 (level 0 of the default pyramid only): the source is the GRAY frame and the level's image -- GaussianBlur with the fixed
// 3 x 3 kernel [1/4 1/2 1/4] (sigma 0), BORDER_REFLECT_101, resize by 1 -- is evaluated where the row loader used to read
// it: I0 is never written nor read (-7 of the 30 bytes per pixel the two kernels moved for this level).  Every intermediate
// of that blur is a multiple of 1/16 below 256, exact in float whatever the order of operations, so the integer form
// (b[x-1] + 2 b[x] + b[x+1] by one v_dot4_u32_u8 on a dword of four neighbouring bytes, rows combined 1-2-1, one conversion,
// one multiplication by 1/16) gives the float expression's bits.  Per-thread constants place the dword inside the row and
// carry the reflected weights at the frame's left / right edge; the rows of the 3-row window roll in registers.
template <int N, class A, bool U8 = false>
__device__ __forceinline__ void polyexp_body(const A& a, const float* __restrict__ img, float* __restrict__ Rbase, int h, int w,
                                             int rows_per_seg, int bx, int by, int z, const uint8_t* __restrict__ gray = nullptr) {
  constexpr int PXW = 3 * PE_RB + 1;   // words per pixel: PE_RB rows x 3 sums + 1 pad
  __shared__ float sv[2][256 * PXW];
  const int tid = threadIdx.x;
  const int np = h * w;
  const float* __restrict__ I = img + (size_t)z * (size_t)np;
  float* __restrict__ R = Rbase + (size_t)z * 5 * (size_t)np;
  const int x = bx * PE_OUT - N + tid;
  const int xc = d_clamp(x, 0, w - 1);
  const int y0 = by * rows_per_seg;
  const int y1 = min(h, y0 + rows_per_seg);
  const bool writer = tid >= N && tid < N + PE_OUT && x < w;
  const unsigned voff = 4u * (unsigned)xc;                 // byte offset of this thread's column in a source row
  const unsigned xo = writer ? (unsigned)x : 0u;           // output column (only writers store)
  // non-writers: beyond any resource's range (the range check covers the vector offset only, not the scalar one).  The row
  // term is added to it for the 16-byte store: with frames of at most 2^26 pixels (polyexp_frame_fits) 0x80000000 + 16 np
  // stays below 2^32 and above the 20 np bytes of the resource, so the sum can neither wrap nor come back into range
  const unsigned xo16 = writer ? 16u * xo : 0x80000000u, xo4 = writer ? 4u * xo : 0x80000000u;
  const __amdgpu_buffer_rsrc_t Rnull = st_rsrc(R, 0);
  const __amdgpu_buffer_rsrc_t Ib = st_rsrc(I, 4 * (size_t)np), Rb = st_rsrc(R, 20 * (size_t)np);
  auto opaque = [](unsigned v) { asm volatile("" : "+v"(v)); return v; };
  auto src = [&](int row) { return bld1(Ib, voff, 4u * (unsigned)w * (unsigned)d_clamp(row, 0, h - 1)); };  // row: uniform
  // U8: dword of gray bytes (lx .. lx + 3) of a row, and the weights that make b[x-1] + 2 b[x] + b[x+1] of it (reflected at the edges)
  const __amdgpu_buffer_rsrc_t Gb = st_rsrc(U8 ? gray + (size_t)z * (size_t)np : nullptr, U8 ? (size_t)np : 0);
  const unsigned lx = !U8 ? 0u : xc == 0 ? 0u : xc >= w - 2 ? (unsigned)(w - 4) : (unsigned)(xc - 1);
  const unsigned wsel = xc == 0 ? 0x00000202u : xc == w - 1 ? 0x02020000u : xc == w - 2 ? 0x01020100u : 0x00010201u;
  auto graw = [&](int grow) { return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(Gb, (int)lx, (int)((unsigned)w * (unsigned)grow), 0); };  // grow: uniform, in range
  auto hsum = [&](unsigned d) { return __builtin_amdgcn_udot4(d, wsel, 0u, false); };
  auto blur16 = [](unsigned a0, unsigned b0, unsigned c0) { return (float)(a0 + 2u * b0 + c0) * 0.0625f; };
  unsigned hsB = 0, hsC = 0;  // U8: row sums of the newest ring row and of the row below it (reflected)

  // ring[j] = source row (y - N + j) of this column for the batch starting at row y
  float ring[2 * N + PE_RB];
  if (!U8) {
#pragma unroll
    for (int j = 0; j < 2 * N + PE_RB; ++j) ring[j] = src(y0 - N + j);
  } else {
    constexpr int NR = 2 * N + PE_RB;
    const int r0 = d_clamp(y0 - N, 0, h - 1);
    const unsigned rm = graw(d_reflect101(r0 - 1, h)), rc = graw(r0);
    unsigned rn[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) rn[j] = graw(d_reflect101(d_clamp(y0 - N + j, 0, h - 1) + 1, h));
    unsigned hsA = hsum(rm);
    hsB = hsum(rc);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      hsC = hsum(rn[j]);
      ring[j] = blur16(hsA, hsB, hsC);
      // the window moves on when the next ring entry is another row (clamped rows repeat)
      if (j + 1 < NR && d_clamp(y0 - N + j + 1, 0, h - 1) != d_clamp(y0 - N + j, 0, h - 1)) { hsA = hsB; hsB = hsC; }
    }
  }

  int buf = 0;
  for (int y = y0; y < y1; y += PE_RB) {
    // vertical pass (float), same association as the scalar reference
#pragma unroll
    for (int r = 0; r < PE_RB; ++r) {
      float r0 = ring[r + N] * a.c.g[0], r1 = 0.f, r2 = 0.f;
#pragma unroll
      for (int k = 1; k <= N; ++k) {
        float p = ring[r + N - k] + ring[r + N + k];
        float t0 = r0 + a.c.g[k] * p;
        float t1 = r2 + a.c.xxg[k] * p;
        p = ring[r + N + k] - ring[r + N - k];
        float t2 = r1 + a.c.xg[k] * p;
        r0 = t0; r1 = t2; r2 = t1;
      }
      float* __restrict__ d = &sv[buf][tid * PXW + 3 * r];
      d[0] = r0; d[1] = r1; d[2] = r2;
    }
    // prefetch the next batch's new source rows while the exchange is in flight
    float nxt[PE_RB];
    unsigned nraw[PE_RB];
#pragma unroll
    for (int r = 0; r < PE_RB; ++r) {
      if (U8) nraw[r] = graw(d_reflect101(min(y + PE_RB + N + r, h - 1) + 1, h));
      else nxt[r] = src(y + PE_RB + N + r);
    }
    __syncthreads();
    // Consume the prefetched rows BEFORE this batch's stores are issued: vmcnt retires in order
    // and counts stores, so waiting for these loads after the stores would also wait for the
    // stores' acknowledgements (the stall that bounded the first version of this kernel).
    if (U8) {
      // rows beyond the frame repeat the last row's value (the float path clamps the row index)
      float last = ring[2 * N + PE_RB - 1];
#pragma unroll
      for (int r = 0; r < PE_RB; ++r) {
        if (y + PE_RB + N + r <= h - 1) {   // uniform
          const unsigned hsA = hsB;
          hsB = hsC; hsC = hsum(nraw[r]);
          last = blur16(hsA, hsB, hsC);
        }
        nxt[r] = last;
      }
    }
#pragma unroll
    for (int j = 0; j < 2 * N; ++j) ring[j] = ring[j + PE_RB];
#pragma unroll
    for (int r = 0; r < PE_RB; ++r) ring[2 * N + r] = nxt[r];
    __builtin_amdgcn_sched_barrier(0);
    {
      // Straight line: every thread runs the horizontal pass and every store instruction is issued on every path --
      // a lane that must not write (halo columns, columns beyond the frame) carries a vector offset beyond the resource's
      // range and a row beyond the segment a null resource, so the hardware drops the store.  With the same eight stores
      // behind the four prefetch loads on every path the compiler's vmcnt for the prefetched rows is exact (it waits
      // for the loads, not for this batch's stores; with a branch around the stores it must assume there were none).
      const unsigned pix0 = opaque(writer ? (unsigned)(tid - N) * PXW : 0u);
#pragma unroll
      for (int r = 0; r < PE_RB; ++r) {
        const float* __restrict__ s = &sv[buf][pix0 + 3 * r];
        float g0 = a.c.g[0];
        const float cx = s[N * PXW], cy = s[N * PXW + 1], cz = s[N * PXW + 2];
        double b1 = cx * g0, b2 = 0, b3 = cy * g0, b4 = 0, b5 = cz * g0, b6 = 0;
#pragma unroll
        for (int k = 1; k <= N; ++k) {
          const float* __restrict__ pp = s + (N + k) * PXW;
          const float* __restrict__ mm = s + (N - k) * PXW;
          const float p0 = pp[0], m0 = mm[0];
          const float p1 = pp[1], m1 = mm[1];
          const float p2 = pp[2], m2 = mm[2];
          double tg = p0 + m0;
          g0 = a.c.g[k];
          b1 = __builtin_fma(tg, (double)g0, b1);
          b4 = __builtin_fma(tg, (double)a.c.xxg[k], b4);
          b2 += (p0 - m0) * a.c.xg[k];
          b3 += (p1 + m1) * g0;
          b6 += (p1 - m1) * a.c.xg[k];
          b5 += (p2 + m2) * g0;
        }
        const bool row_ok = y + r < y1;                                  // uniform
        const unsigned rowo = (unsigned)(y + r) * (unsigned)w;
        bst4(row_ok ? Rb : Rnull, xo16 + 16u * rowo, make_float4((float)(b3 * a.c.ig11), (float)(b2 * a.c.ig11),
                                                                                  (float)(b1 * a.c.ig03 + b5 * a.c.ig33), (float)(b1 * a.c.ig03 + b4 * a.c.ig33)));
        bst1(row_ok ? Rb : Rnull, xo4, 16u * (unsigned)np + 4u * rowo, (float)(b6 * a.c.ig55));
      }
    }
    buf ^= 1;
  }
}

template <int N>
__global__ __launch_bounds__(256) void k_polyexp(PolyArgs a) {
  polyexp_body<N>(a, a.img, a.R, a.h, a.w, a.rows_per_seg, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// level 0 straight from the gray frames (default pyramid: 3 x 3 blur, no resize)
template <int N>
#ifndef ST_PE_U8_WAVES
#define ST_PE_U8_WAVES 1   // experiments: 6 = force the float-source instance's six waves per SIMD (80 registers, 9 spilled: 5.0 ms against 4.5)
#endif
__global__ __launch_bounds__(256, ST_PE_U8_WAVES) void k_polyexp_u8(PolyArgs a) {
  polyexp_body<N, PolyArgs, true>(a, nullptr, a.R, a.h, a.w, a.rows_per_seg, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, a.gray);
}

// Several pyramid levels in ONE launch (small batches: the expansions of the coarse levels are launches of a few dozen
// workgroups each, and every dependent launch costs its ramp and drain): blockIdx.x runs over the (strip, segment)
// pairs of the listed levels, blockIdx.y over the frames.  Same body, so the same bits.
struct PolyLevel {
  const float* img;
  float* R;
  int h, w, rows_per_seg, strips, block0;  // block0: first linear block of this level
};
struct PolyArgsML {
  PolyLevel lv[4];
  int nlv;
  PolyCoef c;
};
template <int N>
__global__ __launch_bounds__(256) void k_polyexp_ml(PolyArgsML a) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < a.nlv && (int)blockIdx.x >= a.lv[i].block0) l = i;
  // select the level's fields with scalar compares (keeps the argument block in scalar registers)
  const float* img = a.lv[0].img; float* R = a.lv[0].R;
  int h = a.lv[0].h, w = a.lv[0].w, rows = a.lv[0].rows_per_seg, strips = a.lv[0].strips, b0 = a.lv[0].block0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (l == i) { img = a.lv[i].img; R = a.lv[i].R; h = a.lv[i].h; w = a.lv[i].w; rows = a.lv[i].rows_per_seg; strips = a.lv[i].strips; b0 = a.lv[i].block0; }
  const int local = (int)blockIdx.x - b0;
  polyexp_body<N>(a, img, R, h, w, rows, local % strips, local / strips, (int)blockIdx.y);
}

// ---------------------------------------------------------------------------------------------
// A5: FarnebackUpdateMatrices for one pixel.
// R layout (one expansion of np = h*w pixels, 20*np bytes): channels 0..3 of OpenCV's interleaved
// 5-channel R as np float4 (16-byte aligned, one wide load per pixel), channel 4 as np floats
// behind them.  Vector-memory instructions issue at ~1 per 20 clocks per CU whatever their width
// (scripts/ubench/vmemrate.hip), and this kernel family is bound by that rate: the 4+1 split
// needs 2 loads for an R0 pixel and 6 for the bilinear 2x2 footprint of R1 instead of the 5 and
// 10 of a five-plane layout, at the same 20 bytes per pixel.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void update_matrices_px(const float* __restrict__ R0, const float* __restrict__ R1,
                                                   int np, int h, int w, int x, int y, float dx, float dy,
                                                   float m[5]) {
  const int o = y * w + x;
  float fx = x + dx, fy = y + dy;
  const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
  float r2, r3, r4, r5, r6;
  fx -= x1; fy -= y1;
  const float4 q = reinterpret_cast<const float4*>(R0)[o];
  const float q0 = q.x, q1 = q.y, q2 = q.z, q3 = q.w, q4 = R0[4 * (size_t)np + o];
  if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) {
    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
    const int gi = y1 * w + x1;
    const float4* __restrict__ Q = reinterpret_cast<const float4*>(R1);
    const float* __restrict__ S = R1 + 4 * (size_t)np;
    const float4 t0 = Q[gi], t1 = Q[gi + 1], b0 = Q[gi + w], b1 = Q[gi + w + 1];
    r2 = a00 * t0.x + a01 * t1.x + a10 * b0.x + a11 * b1.x;
    r3 = a00 * t0.y + a01 * t1.y + a10 * b0.y + a11 * b1.y;
    r4 = a00 * t0.z + a01 * t1.z + a10 * b0.z + a11 * b1.z;
    r5 = a00 * t0.w + a01 * t1.w + a10 * b0.w + a11 * b1.w;
    r6 = a00 * S[gi] + a01 * S[gi + 1] + a10 * S[gi + w] + a11 * S[gi + w + 1];
    r4 = (q2 + r4) * 0.5f;
    r5 = (q3 + r5) * 0.5f;
    r6 = (q4 + r6) * 0.25f;
  } else {
    r2 = r3 = 0.f;
    r4 = q2;
    r5 = q3;
    r6 = q4 * 0.5f;
  }
  r2 = (q0 - r2) * 0.5f;
  r3 = (q1 - r3) * 0.5f;
  r2 += r4 * dy + r6 * dx;
  r3 += r6 * dy + r5 * dx;
  constexpr int BORDER = 5;
  if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) || (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
    // border[] = {0.14, 0.14, 0.4472, 0.4472, 0.4472}, selected with a compare rather than a
    // local array index (keeps it out of scratch)
    auto bsel = [](int i) -> float { return i < 2 ? 0.14f : 0.4472f; };
    const float scale = (x < BORDER ? bsel(x) : 1.f) * (x >= w - BORDER ? bsel(w - x - 1) : 1.f) *
                        (y < BORDER ? bsel(y) : 1.f) * (y >= h - BORDER ? bsel(h - y - 1) : 1.f);
    r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
  }
  m[0] = r4 * r4 + r6 * r6;
  m[1] = (r4 + r5) * r6;
  m[2] = r5 * r5 + r6 * r6;
  m[3] = r4 * r2 + r6 * r3;
  m[4] = r6 * r2 + r5 * r3;
}

// UpdateMatrices split in two so that a thread can put the loads of several rows in flight
// before consuming any (memory-level parallelism): um_issue computes the gather address and
// issues the 2 R0 loads (float4 + float) and the 6 R1 loads (four float4 + two 8-byte pairs of
// channel 4); um_finish does the arithmetic, bit-identical to update_matrices_px.

struct UmLoads {
  float4 q;            // R0 channels 0..3
  float qs;            // R0 channel 4
  float4 t0, t1, b0, b1;  // R1 channels 0..3 at (gi, gi+1, gi+w, gi+w+1)
  f2u ts, bs;          // R1 channel 4 at (gi, gi+1) and (gi+w, gi+w+1)
};

__device__ __forceinline__ void um_gather(const float* __restrict__ R1, unsigned single, unsigned gi, unsigned gb, UmLoads& L);
__device__ __forceinline__ void um_issue(const float* __restrict__ R0, const float* __restrict__ R1, int np, int h,
                                         int w, int x, int y, float2 f, UmLoads& L) {
  float fx = x + f.x, fy = y + f.y;
  const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
  const bool inb = (unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1);
  const unsigned gi = inb ? (unsigned)(y1 * w + x1) : 0u;  // out of range: harmless address, result unused
  const unsigned gb = gi + (inb ? (unsigned)w : 0u);
  const unsigned o = (unsigned)(y * w + x);
  const unsigned single = 16u * (unsigned)np;  // byte offset of the channel-4 plane
  L.q = ldf4(R0, 16u * o);
  L.qs = ldf(R0, single + 4u * o);
  um_gather(R1, single, gi, gb, L);
}

// the six R1 loads of the bilinear footprint; the second column's texel is addressed from a base pointer 16 bytes on
// (uniform arithmetic) rather than by adding 16 to the per-lane offset (a vector instruction per load)
__device__ __forceinline__ void um_gather(const float* __restrict__ R1, unsigned single, unsigned gi, unsigned gb, UmLoads& L) {
  // ST_ABLATE 16 / 64 (experiments, wrong results): what perfect lane sharing (no second-column loads) and a perfect
  // row carry (no top-row loads) would save -- the upper bound of any texel-reuse scheme (profiles/NOTES.md, round 5)
  if (!(ST_ABLATE & 64)) {
    L.t0 = ldf4(R1, 16u * gi);
    if (ST_ABLATE & 16) L.t1 = L.t0; else L.t1 = ldf4(R1 + 4, 16u * gi);
  }
  L.b0 = ldf4(R1, 16u * gb);
  if (ST_ABLATE & 16) L.b1 = L.b0; else L.b1 = ldf4(R1 + 4, 16u * gb);
  if (!(ST_ABLATE & 64)) L.ts = ldf2(R1, single + 4u * gi);
  L.bs = ldf2(R1, single + 4u * gb);
  if (ST_ABLATE & 64) { L.t0 = L.b0; L.t1 = L.b1; L.ts = L.bs; }
}

__device__ __forceinline__ void um_finish(const UmLoads& L, int h, int w, int x, int y, float2 f, float m[5]) {
  // the gather geometry is recomputed from the flow rather than carried in registers across the batch
  const float dx = f.x, dy = f.y;
  float fx = x + dx, fy = y + dy;
  const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
  fx -= x1; fy -= y1;
  const bool inb = (unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1);
  float r2, r3, r4, r5, r6;
  if (inb) {
    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
    r2 = a00 * L.t0.x + a01 * L.t1.x + a10 * L.b0.x + a11 * L.b1.x;
    r3 = a00 * L.t0.y + a01 * L.t1.y + a10 * L.b0.y + a11 * L.b1.y;
    r4 = a00 * L.t0.z + a01 * L.t1.z + a10 * L.b0.z + a11 * L.b1.z;
    r5 = a00 * L.t0.w + a01 * L.t1.w + a10 * L.b0.w + a11 * L.b1.w;
    r6 = a00 * L.ts.x + a01 * L.ts.y + a10 * L.bs.x + a11 * L.bs.y;
    r4 = (L.q.z + r4) * 0.5f;
    r5 = (L.q.w + r5) * 0.5f;
    r6 = (L.qs + r6) * 0.25f;
  } else {
    r2 = r3 = 0.f;
    r4 = L.q.z;
    r5 = L.q.w;
    r6 = L.qs * 0.5f;
  }
  r2 = (L.q.x - r2) * 0.5f;
  r3 = (L.q.y - r3) * 0.5f;
  r2 += r4 * dy + r6 * dx;
  r3 += r6 * dy + r5 * dx;
  constexpr int BORDER = 5;
  if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) || (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
    auto bsel = [](int i) -> float { return i < 2 ? 0.14f : 0.4472f; };
    const float scale = (x < BORDER ? bsel(x) : 1.f) * (x >= w - BORDER ? bsel(w - x - 1) : 1.f) *
                        (y < BORDER ? bsel(y) : 1.f) * (y >= h - BORDER ? bsel(h - y - 1) : 1.f);
    r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
  }
  m[0] = r4 * r4 + r6 * r6;
  m[1] = (r4 + r5) * r6;
  m[2] = r5 * r5 + r6 * r6;
  m[3] = r4 * r2 + r6 * r3;
  m[4] = r6 * r2 + r5 * r3;
}

#if ST_ABLATE & 128
// ST_ABLATE 128 (experiment, wrong results): the ceiling of "R1 rows kept on chip" (round-5 verdict, item 1) -- the six
// per-lane R1 gathers of a pixel replaced by ONE coalesced fill (16 + 4 bytes of the row below, written to a 3-row x
// 272-column window in LDS) and six LDS reads at the gather's column / row offsets.  No barrier orders the fill against
// the reads and the window never tracks the flow: the values are garbage, the instruction mix is the scheme's best case.
// ST_ABLATE 256 adds a workgroup barrier per row batch (what a 3-row window needs to be correct).
constexpr int RW_COLS = 256 + 16;  // B2_T + 16
struct RWin { float4* q; float* s; int xb; };
__device__ __forceinline__ void um_issue_win(const float* __restrict__ R0, const float* __restrict__ R1, int np, int h,
                                             int w, int x, int y, UmLoads& L) {
  const unsigned o = (unsigned)(y * w + x);
  const unsigned single = 16u * (unsigned)np;
  L.q = ldf4(R0, 16u * o);
  L.qs = ldf(R0, single + 4u * o);
  const unsigned of = (unsigned)(min(y + 1, h - 1) * w + x);
  L.t0 = ldf4(R1, 16u * of);
  L.ts.x = ldf(R1, single + 4u * of);
}
__device__ __forceinline__ void um_fill_win(const UmLoads& L, const RWin& win, int slot, int tid) {
  win.q[slot * RW_COLS + tid + 8] = L.t0;
  win.s[slot * RW_COLS + tid + 8] = L.ts.x;
}
__device__ __forceinline__ void um_gather_win(const UmLoads& L, const RWin& win, int slot, int h, int w, int x, int y,
                                              float2 f, UmLoads& G) {
  const float fx = x + f.x, fy = y + f.y;
  const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
  const int c = d_clamp(x1 - win.xb, 0, RW_COLS - 2);
  const int d = d_clamp(y1 - y + 1, 0, 1);
  int s0 = slot + 1 + d; s0 = s0 >= 3 ? s0 - 3 : s0;
  int s1 = s0 + 1; s1 = s1 >= 3 ? s1 - 3 : s1;
  G.q = L.q; G.qs = L.qs;
  const float4* q0 = win.q + s0 * RW_COLS + c;
  const float4* q1 = win.q + s1 * RW_COLS + c;
  G.t0 = q0[0]; G.t1 = q0[1]; G.b0 = q1[0]; G.b1 = q1[1];
  const float* p0 = win.s + s0 * RW_COLS + c;
  const float* p1 = win.s + s1 * RW_COLS + c;
  G.ts.x = p0[0]; G.ts.y = p0[1]; G.bs.x = p1[0]; G.bs.y = p1[1];
}
#endif

// Initial matrices of a level; the flow is zero (coarsest level), a given field, or the
// previous level's flow resized with INTER_LINEAR and multiplied by 1/pyr_scale.
struct UMArgs {
  const float* R;            // frame-indexed expansions of this level (R layout): frame f at R + f*5*np
  const int* pairs;          // device (n_pairs x 2) frame slots, or null => R0 = R, R1 = R1_direct
  const float* R1_direct;
  const float* flow;         // (h,w,2) per pair or null
  const float* coarse;       // (ch,cw,2) per pair or null
  float* M;                  // per pair 5 x np
  int h, w, ch, cw;
  double scale_x, scale_y;   // cw/w, ch/h as cv::resize computes them (1/inv_scale)
  float mul;                 // 1/pyr_scale
};

__device__ __forceinline__ float2 ld_flow(const float* __restrict__ f, int idx) {
  return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(f) + 8u * (unsigned)idx);
}

__global__ __launch_bounds__(256) void k_update_matrices(UMArgs a) {
  const int h = a.h, w = a.w;
  const size_t np = (size_t)h * w;
  const int pr = blockIdx.y;
  const float* R0;
  const float* R1;
  if (a.pairs) {
    R0 = a.R + (size_t)a.pairs[2 * pr] * 5 * np;
    R1 = a.R + (size_t)a.pairs[2 * pr + 1] * 5 * np;
  } else {
    R0 = a.R;
    R1 = a.R1_direct;
  }
  float* __restrict__ M = a.M + (size_t)pr * 5 * np;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / w), x = (int)(i - (size_t)y * w);
    float dx = 0.f, dy = 0.f;
    if (a.coarse) {
      const float* C = a.coarse + (size_t)pr * 2 * a.ch * a.cw;
      // cv::resize INTER_LINEAR, 2 channels: horizontal pass then vertical pass, float
      float fx = (float)((x + 0.5) * a.scale_x - 0.5);
      int sx = (int)floorf(fx);
      fx -= sx;
      if (sx < 0) { fx = 0.f; sx = 0; }
      if (sx >= a.cw - 1) { fx = 0.f; sx = a.cw - 1; }
      float fy = (float)((y + 0.5) * a.scale_y - 0.5);
      int sy = (int)floorf(fy);
      fy -= sy;
      const int ya = d_clamp(sy, 0, a.ch - 1), yb = d_clamp(sy + 1, 0, a.ch - 1);
      const float a1 = fx, a0 = 1.f - fx, b0 = 1.f - fy, b1 = fy;
      float2 ta, tb;
      if (sx + 1 < a.cw) {
        float2 p = ld_flow(C, ya * a.cw + sx), q = ld_flow(C, ya * a.cw + sx + 1);
        ta.x = p.x * a0 + q.x * a1; ta.y = p.y * a0 + q.y * a1;
        p = ld_flow(C, yb * a.cw + sx); q = ld_flow(C, yb * a.cw + sx + 1);
        tb.x = p.x * a0 + q.x * a1; tb.y = p.y * a0 + q.y * a1;
      } else {
        float2 p = ld_flow(C, ya * a.cw + sx);
        ta.x = p.x * 1.f; ta.y = p.y * 1.f;
        p = ld_flow(C, yb * a.cw + sx);
        tb.x = p.x * 1.f; tb.y = p.y * 1.f;
      }
      dx = (ta.x * b0 + tb.x * b1) * a.mul;
      dy = (ta.y * b0 + tb.y * b1) * a.mul;
    } else if (a.flow) {
      float2 f = ld_flow(a.flow + (size_t)pr * 2 * np, (int)i);
      dx = f.x; dy = f.y;
    }
    float m[5];
    update_matrices_px(R0, R1, np, h, w, x, y, dx, dy, m);
    M[i] = m[0]; M[np + i] = m[1]; M[2 * np + i] = m[2]; M[3 * np + i] = m[3]; M[4 * np + i] = m[4];
  }
}

// ---------------------------------------------------------------------------------------------
// A6 (+A5 fused): box blur of M (double running column sums in registers, row exchange through
// LDS, direct horizontal window sum in double), 2x2 solve, then either the flow store (last
// iteration) or UpdateMatrices with the fresh flow into the other M buffer.
// ---------------------------------------------------------------------------------------------
struct BlurArgs {
  const float* R;        // frame-indexed expansions (see UMArgs), or direct R0/R1
  const int* pairs;
  const float* R1_direct;
  const float* Min;      // per pair 5 x np
  float* Mout;           // per pair 5 x np (update) -- never aliases Min
  float* flow;           // per pair (h,w,2) contiguous, or
  float* const* flow_ptrs;  // device table of per-pair output frames (used when non-null)
  int h, w, rows_per_seg, m;  // m = block_size/2
  int update, write_flow;
  double scale;          // 1/(block_size^2)
};

constexpr int BLUR_T = 256;

__global__ __launch_bounds__(BLUR_T) void k_blur_update(BlurArgs a) {
  __shared__ double sv[2][5][BLUR_T];
  const int tid = threadIdx.x;
  const int h = a.h, w = a.w, m = a.m;
  const size_t np = (size_t)h * w;
  const int pr = blockIdx.z;
  const float* __restrict__ Min = a.Min + (size_t)pr * 5 * np;
  const int x = (int)blockIdx.x * (BLUR_T - 2 * m) - m + tid;
  const int xc = d_clamp(x, 0, w - 1);
  const int y0 = blockIdx.y * a.rows_per_seg;
  const int y1 = min(h, y0 + a.rows_per_seg);
  const bool writer = tid >= m && tid < BLUR_T - m && x < w;

  const float* R0 = nullptr;
  const float* R1 = nullptr;
  float* Mout = nullptr;
  if (a.update) {
    if (a.pairs) {
      R0 = a.R + (size_t)a.pairs[2 * pr] * 5 * np;
      R1 = a.R + (size_t)a.pairs[2 * pr + 1] * 5 * np;
    } else {
      R0 = a.R;
      R1 = a.R1_direct;
    }
    Mout = a.Mout + (size_t)pr * 5 * np;
  }
  float* flow = nullptr;
  if (a.write_flow) flow = a.flow_ptrs ? st_gl(a.flow_ptrs[pr]) : a.flow + (size_t)pr * 2 * np;

  // window sum for row y0: rows y0-m .. y0+m with replicated borders
  double vs[5] = {0, 0, 0, 0, 0};
  if (y0 == 0) {
    // the top segment reproduces the reference's initialisation order exactly:
    // float(M[0]*(m+2)) + sum_{1..m-1} M[y] + float(M[m] - M[0])
#pragma unroll
    for (int c = 0; c < 5; ++c) vs[c] = (double)(Min[c * np + xc] * (float)(m + 2));
    for (int yy = 1; yy < m; ++yy) {
      const size_t o = (size_t)min(yy, h - 1) * w + xc;
#pragma unroll
      for (int c = 0; c < 5; ++c) vs[c] += (double)Min[c * np + o];
    }
    const size_t oa = (size_t)min(m, h - 1) * w + xc;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const float d = Min[c * np + oa] - Min[c * np + xc];
      vs[c] += d;
    }
  } else {
    for (int dy = -m; dy <= m; ++dy) {
      const size_t o = (size_t)d_clamp(y0 + dy, 0, h - 1) * w + xc;
#pragma unroll
      for (int c = 0; c < 5; ++c) vs[c] += (double)Min[c * np + o];
    }
  }

  for (int y = y0; y < y1; ++y) {
    double(*s)[BLUR_T] = sv[y & 1];
#pragma unroll
    for (int c = 0; c < 5; ++c) s[c][tid] = vs[c];
    // slide the window for the next row: += float(M[y+1+m] - M[y-m])
    if (y + 1 < y1) {
      const size_t oa = (size_t)d_clamp(y + 1 + m, 0, h - 1) * w + xc;
      const size_t ob = (size_t)d_clamp(y - m, 0, h - 1) * w + xc;
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const float d = Min[c * np + oa] - Min[c * np + ob];
        vs[c] += d;
      }
    }
    __syncthreads();
    if (writer) {
      double g[5];
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        double t = s[c][tid - m];
        for (int k = -m + 1; k <= m; ++k) t += s[c][tid + k];
        g[c] = t * a.scale;
      }
      const double idet = 1. / (g[0] * g[2] - g[1] * g[1] + 1e-3);
      const float u = (float)((g[0] * g[4] - g[1] * g[3]) * idet);
      const float v = (float)((g[2] * g[3] - g[1] * g[4]) * idet);
      const size_t o = (size_t)y * w + x;
      if (flow) *reinterpret_cast<float2*>(flow + 2 * o) = make_float2(u, v);
      if (a.update) {
        float mm[5];
        update_matrices_px(R0, R1, np, h, w, x, y, u, v, mm);
#pragma unroll
        for (int c = 0; c < 5; ++c) Mout[c * np + o] = mm[c];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fast path of A6(+A5) for block_size = 2*M+1 with M <= 8 (the reference uses 15, M = 7).
//
// One workgroup owns a 256-column strip (240 outputs + 8 halo columns each side) of one pair
// and marches down a vertical segment in batches of RB = 5 rows with two barriers per batch:
//   phase 1  thread = column.  The 2M+1 source rows of the vertical window live in a register
//            ring (statically indexed: the batch loop is unrolled over the ring period), so each
//            M value is read from HBM exactly once; the running column sum is kept in double
//            exactly like the reference (+= float(M[y+m] - M[y-m-1])).  The five sums of each
//            batch row go to LDS (float).
//   phase 2  thread = (row, 5-pixel segment).  Sliding-window horizontal sum in double over the
//            LDS row (19 reads for 5 outputs instead of 75), 2x2 solve, flow -> LDS.
//   phase 3  thread = column again: coalesced flow store (last iteration) or UpdateMatrices
//            (R0 stream, bilinear R1 gather, M' stream) with the fresh flow.
// LDS: 25.6 KB sums + 10 KB flow per workgroup -> 4 workgroups per CU.
// ---------------------------------------------------------------------------------------------
constexpr int B2_T = 256, B2_HALO = 8, B2_OUT = B2_T - 2 * B2_HALO, B2_RB = 5, B2_SEG = 5;
constexpr int B2_NSEG = B2_OUT / B2_SEG;  // 48 segments per row

template <int M, bool RING, int U3, int WAVES>
__global__ __launch_bounds__(B2_T, WAVES) void k_blur_update_v2(BlurArgs a) {
  constexpr int W = 2 * M + 1;
  static_assert(W % B2_RB == 0 && M <= B2_HALO, "ring period must be a multiple of the batch");
  __shared__ float V[B2_RB][5][B2_T];
  __shared__ float2 F[B2_RB][B2_T];
  const int tid = threadIdx.x;
  const int h = a.h, w = a.w;
  const int np = h * w;  // 32-bit element offsets: one VGPR per address, SGPR base
  const int pr = blockIdx.z;
  const float* __restrict__ Min = a.Min + (size_t)pr * 5 * (size_t)np;
  const int x = (int)blockIdx.x * B2_OUT - B2_HALO + tid;
  const int xc = d_clamp(x, 0, w - 1);
  const int y0 = blockIdx.y * a.rows_per_seg;
  const int y1 = min(h, y0 + a.rows_per_seg);
  const bool writer = tid >= B2_HALO && tid < B2_T - B2_HALO && x < w;

  const float* R0 = nullptr;
  const float* R1 = nullptr;
  float* Mout = nullptr;
  if (a.update) {
    if (a.pairs) {
      R0 = a.R + (size_t)a.pairs[2 * pr] * 5 * (size_t)np;
      R1 = a.R + (size_t)a.pairs[2 * pr + 1] * 5 * (size_t)np;
    } else {
      R0 = a.R;
      R1 = a.R1_direct;
    }
    Mout = a.Mout + (size_t)pr * 5 * (size_t)np;
  }
  float* flow = nullptr;
  if (a.write_flow) flow = a.flow_ptrs ? st_gl(a.flow_ptrs[pr]) : a.flow + (size_t)pr * 2 * (size_t)np;

  // ring slot s holds source row y0 - M + s (clamped) at entry; vs = window sum of row y0
  float ring[RING ? W : 1][5];
  double vs[5];
  if (!RING) {
    // no register ring: the row leaving the window is re-read (served by L2 / Infinity Cache / HBM)
    if (y0 == 0) {
#pragma unroll
      for (int c = 0; c < 5; ++c) vs[c] = (double)(Min[c * np + xc] * (float)(M + 2));
      for (int yy = 1; yy < M; ++yy) {
        const int o = min(yy, h - 1) * w + xc;
#pragma unroll
        for (int c = 0; c < 5; ++c) vs[c] += (double)Min[c * np + o];
      }
      const int oa = min(M, h - 1) * w + xc;
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const float d = Min[c * np + oa] - Min[c * np + xc];
        vs[c] += d;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 5; ++c) vs[c] = 0;
      for (int dy = -M; dy <= M; ++dy) {
        const int o = d_clamp(y0 + dy, 0, h - 1) * w + xc;
#pragma unroll
        for (int c = 0; c < 5; ++c) vs[c] += (double)Min[c * np + o];
      }
    }
  } else {
#pragma unroll
  for (int s = 0; s < W; ++s) {
    const int o = d_clamp(y0 - M + s, 0, h - 1) * w + xc;
#pragma unroll
    for (int c = 0; c < 5; ++c) ring[s][c] = Min[c * np + o];
  }
  if (y0 == 0) {
    // reference initialisation order: float(M[0]*(m+2)) + sum_{1..m-1} M[y] + float(M[m] - M[0])
    // (ring slots M..2M hold rows 0..M here; slots 0..M-1 hold row 0 as well)
#pragma unroll
    for (int c = 0; c < 5; ++c) vs[c] = (double)(ring[M][c] * (float)(M + 2));
#pragma unroll
    for (int yy = 1; yy < M; ++yy)
#pragma unroll
      for (int c = 0; c < 5; ++c) vs[c] += (double)ring[M + yy][c];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const float d = ring[2 * M][c] - ring[M][c];
      vs[c] += d;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 5; ++c) vs[c] = 0;
#pragma unroll
    for (int s = 0; s < W; ++s)
#pragma unroll
      for (int c = 0; c < 5; ++c) vs[c] += (double)ring[s][c];
  }
  }

  for (int yb = y0; yb < y1; yb += W) {
#pragma unroll
    for (int b = 0; b < W / B2_RB; ++b) {
      const int ybb = yb + b * B2_RB;
      if (ybb < y1) {  // workgroup-uniform
        // ---- phase 1: vertical window ----
        float nw[B2_RB][5], od[RING ? 1 : B2_RB][5];
#pragma unroll
        for (int r = 0; r < B2_RB; ++r) {
          const int o = d_clamp(ybb + r + M + 1, 0, h - 1) * w + xc;
#pragma unroll
          for (int c = 0; c < 5; ++c) nw[r][c] = Min[c * np + o];
          if (!RING) {
            const int oo = d_clamp(ybb + r - M, 0, h - 1) * w + xc;
#pragma unroll
            for (int c = 0; c < 5; ++c) od[r][c] = Min[c * np + oo];
          }
        }
#pragma unroll
        for (int r = 0; r < B2_RB; ++r) {
#pragma unroll
          for (int c = 0; c < 5; ++c) {
            V[r][c][tid] = (float)vs[c];
            const float old = RING ? ring[RING ? b * B2_RB + r : 0][c] : od[RING ? 0 : r][c];
            const float d = nw[r][c] - old;
            vs[c] += d;
            if (RING) ring[RING ? b * B2_RB + r : 0][c] = nw[r][c];
          }
        }
        __syncthreads();
        // ---- phase 2: horizontal window + solve ----
        if (tid < B2_RB * B2_NSEG) {
          const int r = tid / B2_NSEG, sg = tid - r * B2_NSEG;
          const int j0 = B2_HALO + sg * B2_SEG;
          // five running window sums (one per channel) slide together along the segment
          double t[5];
#pragma unroll
          for (int c = 0; c < 5; ++c) {
            const float* vp = &V[r][c][j0 - M];
            double acc = vp[0];
#pragma unroll
            for (int i = 1; i < W; ++i) acc += (double)vp[i];
            t[c] = acc;
            __builtin_amdgcn_sched_barrier(0);  // keep one channel's 15 LDS reads live at a time
          }
#pragma unroll
          for (int i = 0; i < B2_SEG; ++i) {
            if (i > 0) {
#pragma unroll
              for (int c = 0; c < 5; ++c) t[c] += (double)V[r][c][j0 + i + M] - (double)V[r][c][j0 + i - M - 1];
            }
            const double g11 = t[0] * a.scale, g12 = t[1] * a.scale, g22 = t[2] * a.scale;
            const double h1 = t[3] * a.scale, h2 = t[4] * a.scale;
            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
            F[r][j0 + i] = make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __syncthreads();
        // ---- phase 3: flow store / UpdateMatrices ----
        if (writer) {
#pragma unroll U3
          for (int r = 0; r < B2_RB; ++r) {
            const int y = ybb + r;
            if (y < y1) {
              const float2 f = F[r][tid];
              const int o = y * w + x;
              if (flow) *reinterpret_cast<float2*>(flow + 2 * o) = f;
              if (a.update) {
                float mm[5];
                update_matrices_px(R0, R1, np, h, w, x, y, f.x, f.y, mm);
#pragma unroll
                for (int c = 0; c < 5; ++c) Mout[c * np + o] = mm[c];
              }
            }
          }
        }
      }
    }
  }
}

// ST_ABLATE (build-time bit mask, experiments only -- results are then meaningless): 4 = k_flow_iter3
// without its expansion loads.  Timing a build with a part removed shows what that part costs in place
// (profiles/README.md lists the round-2 ablations of the previous kernel generation).

#ifndef ST_EXP_D
#define ST_EXP_D 1  // gather queue depth of k_flow_iter3 (experiments)
#endif

// ---------------------------------------------------------------------------------------------
// One full Farneback iteration without materialising M:
//     flow_out = solve(box_15x15(UpdateMatrices(R0, R1, flow_in)))
// Same three-phase structure as k_blur_update_v2, but the row that enters the vertical window is
// COMPUTED in phase 1 from (flow_in, R0, R1) instead of being loaded, and phase 3 only stores the
// new flow.  HBM traffic per pixel and iteration: flow in 8 + R0 20 + R1 20 (gather) + flow out 8
// = 56 B, against 60 (UpdateMatrices) + 20 + 8 for the unfused pair of stages; M (20 B/px written
// and read back every iteration in the reference's formulation) never exists in memory.
// flow_in is (a) zero (coarsest level, first iteration), (b) resize(coarse flow)*(1/pyr_scale)
// evaluated on the fly (first iteration of a finer level) or (c) the previous iteration's flow.
// The values of M are bit-identical to the materialised ones, so results do not change.
// ---------------------------------------------------------------------------------------------
struct IterArgs {
  const float* R;           // frame-indexed expansions of this level (R layout), or direct R0
  const int* pairs;         // device (n_pairs x 2) frame slots, or null => R0 = R, R1 = R1_direct
  const float* R1_direct;
  const float* flow_in;     // per pair (h,w,2) or null
  const float* coarse;      // per pair (ch,cw,2) or null
  float* flow_out;          // per pair (h,w,2), or
  float* const* flow_ptrs;  // device table of per-pair output frames (used when non-null)
  int h, w, ch, cw, rows_per_seg;
  int out_w;                // k_flow_iter_roles: output columns per strip (multiple of 8)
  double scale_x, scale_y;  // coarse/fine size ratios as cv::resize computes them
  float mul;                // 1/pyr_scale
  double scale;             // 1/(block_size^2)
};

enum { FLOW_ZERO = 0, FLOW_FIELD = 1, FLOW_COARSE = 2, FLOW_ANY = 3,
       FLOW_COARSE2 = 4 };  // FLOW_COARSE2: level transition with the coarse level exactly half as tall

// x-dependent half of the INTER_LINEAR up-sample (fixed per thread: a thread owns one column)
struct CoarseX {
  int sx;
  float a0, a1;
  bool pair;  // sx+1 is inside the coarse row
};

__device__ __forceinline__ CoarseX coarse_x(const IterArgs& a, int x) {
  CoarseX cx;
  float fx = (float)((x + 0.5) * a.scale_x - 0.5);
  int sx = (int)floorf(fx);
  fx -= sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= a.cw - 1) { fx = 0.f; sx = a.cw - 1; }
  cx.sx = sx; cx.a1 = fx; cx.a0 = 1.f - fx; cx.pair = sx + 1 < a.cw;
  return cx;
}

// Fetch of the input flow vector of pixel (x, y), split in two so that the fused iteration can
// request it a batch ahead and do the up-sampling arithmetic only when the data has arrived:
// flow_issue puts the loads in flight (the field's vector, or the 2x2 coarse neighbourhood as two
// 16-byte loads), flow_finish turns them into the vector.
typedef float f4u8 __attribute__((ext_vector_type(4), aligned(8)));
struct FlowRaw {
  f4u8 pa, pb;  // coarse rows ya / yb at columns (sx, sx+1); field mode: pa.xy is the vector
};

template <int MODE = FLOW_ANY>
__device__ __forceinline__ void flow_issue(const IterArgs& a, const float* __restrict__ fin,
                                           const float* __restrict__ C, const CoarseX& cx, int x, int y, FlowRaw& r) {
  if (MODE == FLOW_COARSE || MODE == FLOW_COARSE2 || (MODE == FLOW_ANY && C)) {
    const float fy = (float)((y + 0.5) * a.scale_y - 0.5);
    const int sy = (int)floorf(fy);
    const int ya = d_clamp(sy, 0, a.ch - 1), yb = d_clamp(sy + 1, 0, a.ch - 1);
    if (cx.pair) {
      // the two horizontally adjacent coarse vectors in ONE 16-byte load (8-byte aligned)
      r.pa = *reinterpret_cast<const f4u8*>(reinterpret_cast<const char*>(C) + 8u * (unsigned)(ya * a.cw + cx.sx));
      r.pb = *reinterpret_cast<const f4u8*>(reinterpret_cast<const char*>(C) + 8u * (unsigned)(yb * a.cw + cx.sx));
    } else {
      float2 p = ld_flow(C, ya * a.cw + cx.sx);
      r.pa.x = p.x; r.pa.y = p.y; r.pa.z = 0.f; r.pa.w = 0.f;
      p = ld_flow(C, yb * a.cw + cx.sx);
      r.pb.x = p.x; r.pb.y = p.y; r.pb.z = 0.f; r.pb.w = 0.f;
    }
  } else if (MODE == FLOW_FIELD || (MODE == FLOW_ANY && fin)) {
    const float2 p = ld_flow(fin, y * a.w + x);
    r.pa.x = p.x; r.pa.y = p.y;
  }
}

template <int MODE = FLOW_ANY>
__device__ __forceinline__ float2 flow_finish(const IterArgs& a, const float* __restrict__ fin,
                                              const float* __restrict__ C, const CoarseX& cx, int y, const FlowRaw& r) {
  if (MODE == FLOW_COARSE || MODE == FLOW_COARSE2 || (MODE == FLOW_ANY && C)) {
    // cv::resize INTER_LINEAR, 2 channels: horizontal pass then vertical pass, float
    float fy = (float)((y + 0.5) * a.scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    const float a1 = cx.a1, a0 = cx.a0, b0 = 1.f - fy, b1 = fy;
    float2 ta, tb;
    if (cx.pair) {
      ta.x = r.pa.x * a0 + r.pa.z * a1; ta.y = r.pa.y * a0 + r.pa.w * a1;
      tb.x = r.pb.x * a0 + r.pb.z * a1; tb.y = r.pb.y * a0 + r.pb.w * a1;
    } else {
      ta.x = r.pa.x * 1.f; ta.y = r.pa.y * 1.f;
      tb.x = r.pb.x * 1.f; tb.y = r.pb.y * 1.f;
    }
    return make_float2((ta.x * b0 + tb.x * b1) * a.mul, (ta.y * b0 + tb.y * b1) * a.mul);
  }
  if (MODE == FLOW_FIELD || (MODE == FLOW_ANY && fin)) return make_float2(r.pa.x, r.pa.y);
  return make_float2(0.f, 0.f);
}

// row index of the first coarse row a fine row y0 interpolates from (level transitions)
__device__ __forceinline__ int coarse_row0(const IterArgs& a, int y0) {
  return (int)floorf((float)((y0 + 0.5) * a.scale_y - 0.5));
}

template <int MODE = FLOW_ANY>
__device__ __forceinline__ float2 iter_flow_at(const IterArgs& a, const float* __restrict__ fin,
                                               const float* __restrict__ C, const CoarseX& cx, int x, int y) {
  FlowRaw r;
  flow_issue<MODE>(a, fin, C, cx, x, y, r);
  return flow_finish<MODE>(a, fin, C, cx, y, r);
}

// ---------------------------------------------------------------------------------------------
// k_flow_iter3: the marching iteration restructured around what the round-2 measurements showed: the
// kernel's time follows its VALU instruction count (two waves per SIMD, ~60 % of the time issuing
// VALU), and 63 of the 112 double-precision instructions per pixel were the horizontal window sums,
// restarted every 3 pixels.  Here
//   * phase 2 runs once per GROUP = 8 rows on (row, 8-pixel segment) threads: one fresh 15-term sum and
//     seven slides per thread and channel, 36 instead of 63 double-precision instructions per pixel,
//     and two barriers per 8 rows instead of per 3;
//   * the ring has 16 statically indexed slots (the entering row overwrites the row that left one step
//     earlier), so a 16-row period = two groups is unrolled and nothing is shifted, for every flow
//     source;
//   * phase 1 runs GROUP / RB batches of RB rows back to back between two barriers (RB = 4 for the zero
//     / field sources, 2 for the level transitions, whose up-sampling needs the registers);
//   * the column sums are re-anchored every 32 rows (fresh sum of the ring in row order).
// LDS rows are stored with one pad word per 8 columns: phase 2's lanes are 8 columns apart, 9 words with
// the padding, which keeps their reads conflict-free.
// Canonical association (shared with k_flow_iter_tile): vertical anchors at rows 32 j, horizontal
// anchors at columns 8 i.
// ---------------------------------------------------------------------------------------------
constexpr int F3_GROUP = 8, F3_SW = 8, F3_NSEG = B2_OUT / F3_SW, F3_RING = 16, F3_ANCHOR = 32;
constexpr int F3_PADW = B2_T + B2_T / 8;
static_assert(B2_OUT % F3_SW == 0 && F3_GROUP * F3_NSEG <= B2_T && F3_ANCHOR % F3_RING == 0 && F3_RING % F3_GROUP == 0, "bad v3 geometry");
__device__ __forceinline__ constexpr int f3_pos(int x) { return x + (x >> 3); }

// Exactly half as tall (a.h == 2 a.ch, the only geometry FLOW_COARSE2 is launched for): scale_y is exactly 0.5, so
// fy = (y + 0.5) 0.5 - 0.5 = y / 2 - 0.25 exactly -- even rows: sy = y / 2 - 1, weight 0.75; odd rows: sy = (y - 1) / 2,
// weight 0.25 -- and the row geometry is integer arithmetic on wave-uniform values (scalar unit) instead of ~20 double /
// float vector instructions per row.  The same numbers as the generic expressions, which are exact here.
__device__ __forceinline__ int coarse2_row(int y) { return ((y + 1) >> 1) - 1; }

// coarse rows s .. s + RB/2 + 1 cover the RB fine rows of a batch when the coarse level is exactly half as tall
template <int RB>
struct FlowRawN {
  f4u8 row[RB / 2 + 2];
};
template <int RB>
__device__ __forceinline__ void coarseN_issue(const IterArgs& a, const float* __restrict__ C, const CoarseX& cx, int y0,
                                              FlowRawN<RB>& r) {
  const int s = coarse2_row(y0);  // == coarse_row0(a, y0) for the exact-half geometry this is used for
#pragma unroll
  for (int j = 0; j < RB / 2 + 2; ++j) {
    const int yy = d_clamp(s + j, 0, a.ch - 1);
    if (cx.pair) {
      r.row[j] = *reinterpret_cast<const f4u8*>(reinterpret_cast<const char*>(C) + 8u * (unsigned)(yy * a.cw + cx.sx));
    } else {
      const float2 p = ld_flow(C, yy * a.cw + cx.sx);
      r.row[j].x = p.x; r.row[j].y = p.y; r.row[j].z = 0.f; r.row[j].w = 0.f;
    }
  }
}
// A batch's RB flow vectors from its shared coarse rows: each coarse row interpolated horizontally ONCE (the generic path
// does it per fine row), then combined vertically with the two weights; operands and operations per vector are those of
// coarseN_finish, so the bits are.
template <int RB>
__device__ __forceinline__ void coarse2_finish_batch(const IterArgs& a, const CoarseX& cx, int y0, const int (&ys)[RB],
                                                     const FlowRawN<RB>& r, float2 (&out)[RB]) {
  const int s = coarse2_row(y0);
  const float a1 = cx.a1, a0 = cx.a0;
  float2 hx[RB / 2 + 2];
#pragma unroll
  for (int j = 0; j < RB / 2 + 2; ++j) {
    if (cx.pair) {
      hx[j].x = r.row[j].x * a0 + r.row[j].z * a1; hx[j].y = r.row[j].y * a0 + r.row[j].w * a1;
    } else {
      hx[j].x = r.row[j].x * 1.f; hx[j].y = r.row[j].y * 1.f;
    }
  }
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int y = ys[i];
    const int idx = coarse2_row(y) - s;  // 0 .. RB / 2, wave-uniform
    const float fy = (y & 1) ? 0.25f : 0.75f;
    const float b0 = 1.f - fy, b1 = fy;
    float2 ta = hx[0], tb = hx[1];
#pragma unroll
    for (int j = 1; j <= RB / 2; ++j)
      if (idx == j) { ta = hx[j]; tb = hx[j + 1]; }
    out[i] = make_float2((ta.x * b0 + tb.x * b1) * a.mul, (ta.y * b0 + tb.y * b1) * a.mul);
  }
}

template <int RB>
__device__ __forceinline__ float2 coarseN_finish(const IterArgs& a, const CoarseX& cx, int y0, int y, const FlowRawN<RB>& r) {
  float fy = (float)((y + 0.5) * a.scale_y - 0.5);
  const int sy = (int)floorf(fy);
  fy -= sy;
  const int idx = sy - coarse_row0(a, y0);  // 0 .. RB/2
  f4u8 pa = r.row[0], pb = r.row[1];
#pragma unroll
  for (int j = 1; j <= RB / 2; ++j)
    if (idx == j) { pa = r.row[j]; pb = r.row[j + 1]; }
  const float a1 = cx.a1, a0 = cx.a0, b0 = 1.f - fy, b1 = fy;
  float2 ta, tb;
  if (cx.pair) {
    ta.x = pa.x * a0 + pa.z * a1; ta.y = pa.y * a0 + pa.w * a1;
    tb.x = pb.x * a0 + pb.z * a1; tb.y = pb.y * a0 + pb.w * a1;
  } else {
    ta.x = pa.x * 1.f; ta.y = pa.y * 1.f;
    tb.x = pb.x * 1.f; tb.y = pb.y * 1.f;
  }
  return make_float2((ta.x * b0 + tb.x * b1) * a.mul, (ta.y * b0 + tb.y * b1) * a.mul);
}

// Workgroups are handed to the 8 XCDs round-robin by linear id, and consecutive pairs share a frame's expansion
// (R1 of pair p is R0 of pair p + 1): the sharing reaches the XCD's L2 only when the same (strip, segment) column of
// consecutive pairs runs on ONE XCD at about the same time.  With 8 k columns that is what the natural order gives
// (column c -> XCD c mod 8); with 1, 2 or 4 columns (the coarser levels) consecutive pairs land on different XCDs.
// This remap gives XCD j the column j mod C and the pairs of group j / C, consecutively.  Pure scheduling
// (measured: 1.786 -> 1.775 ms per launch on average over the 12 launches of a 256-pair 1080p step).
__device__ __forceinline__ void xcd_remap(unsigned& bx, unsigned& by, unsigned& bz) {
  const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
  const unsigned C = gx * gy, N = C * gz;
  if (N % 8u) return;
  const unsigned L = bx + gx * (by + gy * bz), xcd = L % 8u, slot = L / 8u;
  unsigned col, pair;
  if (C % 8u == 0) {
    const unsigned cpx = C / 8u;
    col = xcd + 8u * (slot % cpx);
    pair = slot / cpx;
  } else if (8u % C == 0 && gz % (8u / C) == 0) {
    col = xcd % C;
    pair = (xcd / C) * (gz / (8u / C)) + slot;
  } else {
    return;
  }
  bx = col % gx; by = col / gx; bz = pair;
}

template <int M, int RB, int MODE, int D>
__global__ __launch_bounds__(B2_T, 2) void k_flow_iter3(IterArgs a) {
  typedef float f2v __attribute__((ext_vector_type(2)));
  constexpr int W = 2 * M + 1;
  static_assert((F3_RING / RB) % D == 0, "the gather queue must rotate a whole number of times per ring period");
  static_assert(W == F3_RING - 1 && F3_GROUP % RB == 0 && M <= B2_HALO, "ring of 16 = window of 15 + the entering row");
  __shared__ float Vs[F3_GROUP][5][F3_PADW];
  __shared__ float2 Fs[F3_GROUP][F3_PADW];
#if ST_ABLATE & 128
  __shared__ float4 Rw4[3 * RW_COLS];
  __shared__ float Rw1[3 * RW_COLS];
  int wslot = 0;
#endif
  const int tid = threadIdx.x;
  const int h = a.h, w = a.w;
  const int np = h * w;
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  xcd_remap(bx, by, bz);
  const int pr = bz;
  const int x = (int)bx * B2_OUT - B2_HALO + tid;
  const int xc = d_clamp(x, 0, w - 1);
  const int y0 = by * a.rows_per_seg;
  const int y1 = min(h, y0 + a.rows_per_seg);
  const bool writer = tid >= B2_HALO && tid < B2_T - B2_HALO && x < w;
  const int vpos = f3_pos(tid);
#if ST_ABLATE & 128
  const RWin win{Rw4, Rw1, (int)bx * B2_OUT - B2_HALO - 8};
#endif
  // paired flow stores (even widths): writer k = tid - HALO stores pixels (2 j, 2 j + 1) of rows 4 hh .. 4 hh + 3 of a group,
  // j = k mod (OUT / 2), hh = k / (OUT / 2)
  typedef float f4v __attribute__((ext_vector_type(4), aligned(8)));
  const bool pair_tid = tid >= B2_HALO && tid < B2_T - B2_HALO;
  const int pair_k = pair_tid ? tid - B2_HALO : 0, pair_j = pair_k % (B2_OUT / 2);
  const int pair_r0 = (pair_k / (B2_OUT / 2)) * (F3_GROUP / 2);
  const int pair_x = (int)bx * B2_OUT + 2 * pair_j;
  const int pair_pos = f3_pos(B2_HALO + 2 * pair_j);

  const float* __restrict__ R0;
  const float* __restrict__ R1;
  if (a.pairs) {
    R0 = a.R + (size_t)a.pairs[2 * pr] * 5 * (size_t)np;
    R1 = a.R + (size_t)a.pairs[2 * pr + 1] * 5 * (size_t)np;
  } else {
    R0 = a.R;
    R1 = a.R1_direct;
  }
  const float* __restrict__ fin = a.flow_in ? a.flow_in + (size_t)pr * 2 * (size_t)np : nullptr;
  const float* __restrict__ C = a.coarse ? a.coarse + (size_t)pr * 2 * (size_t)a.ch * a.cw : nullptr;
  float* fout = a.flow_ptrs ? st_gl(a.flow_ptrs[pr]) : a.flow_out + (size_t)pr * 2 * (size_t)np;
  const CoarseX cx = (MODE == FLOW_COARSE || MODE == FLOW_COARSE2) ? coarse_x(a, xc) : CoarseX{0, 1.f, 0.f, false};

  // ring slot s holds M of source row y0 - M + s (clamped), s = 0 .. 14; slot 15 takes the first entering row
  float ring[F3_RING][5];
#pragma unroll
  for (int s0 = 0; s0 < W; s0 += 3) {
    UmLoads Li[3];
    float2 f[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int yy = d_clamp(y0 - M + s0 + i, 0, h - 1);
      f[i] = iter_flow_at<MODE>(a, fin, C, cx, xc, yy);
      um_issue(R0, R1, np, h, w, xc, yy, f[i], Li[i]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) um_finish(Li[i], h, w, xc, d_clamp(y0 - M + s0 + i, 0, h - 1), f[i], ring[s0 + i]);
    __builtin_amdgcn_sched_barrier(0);  // bound the loads kept in flight
  }
#pragma unroll
  for (int c = 0; c < 5; ++c) ring[F3_RING - 1][c] = 0.f;
  double vs[5];
  if (y0 == 0) {
    // reference initialisation order: float(M[0]*(m+2)) + sum_{1..m-1} M[y] + float(M[m] - M[0])
#pragma unroll
    for (int c = 0; c < 5; ++c) vs[c] = (double)(ring[M][c] * (float)(M + 2));
#pragma unroll
    for (int yy = 1; yy < M; ++yy)
#pragma unroll
      for (int c = 0; c < 5; ++c) vs[c] += (double)ring[M + yy][c];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const float d = ring[2 * M][c] - ring[M][c];
      vs[c] += d;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 5; ++c) vs[c] = 0;
#pragma unroll
    for (int s = 0; s < W; ++s)
#pragma unroll
      for (int c = 0; c < 5; ++c) vs[c] += (double)ring[s][c];
  }

  // software pipeline over batches of RB rows: L is a queue of D batches of gathers in flight (fcur their
  // flows), fnext the flows of the batch that enters the queue next, raw the flow loads of the batch after it.
  // (D = 2 -- four rows of gathers in flight -- would hide more of the load latency, which is a third of
  // this kernel's time, but does not fit: 133 spilled registers for the field source.  D = 1 everywhere.)
  float2 fcur[D][RB], fnext[RB];
  UmLoads L[D][RB];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      fcur[d][r] = iter_flow_at<MODE>(a, fin, C, cx, xc, d_clamp(y0 + d * RB + r + M + 1, 0, h - 1));
      um_issue(R0, R1, np, h, w, xc, d_clamp(y0 + d * RB + r + M + 1, 0, h - 1), fcur[d][r], L[d][r]);
    }
#pragma unroll
  for (int r = 0; r < RB; ++r) fnext[r] = iter_flow_at<MODE>(a, fin, C, cx, xc, d_clamp(y0 + D * RB + r + M + 1, 0, h - 1));
  FlowRaw raw[MODE == FLOW_COARSE2 ? 1 : RB];
  FlowRawN<RB> rawn;
  if (MODE == FLOW_COARSE2) {
    coarseN_issue<RB>(a, C, cx, d_clamp(y0 + (D + 1) * RB + M + 1, 0, h - 1), rawn);
  } else {
#pragma unroll
    for (int r = 0; r < RB; ++r) flow_issue<MODE>(a, fin, C, cx, xc, d_clamp(y0 + (D + 1) * RB + r + M + 1, 0, h - 1), raw[r]);
  }

#pragma unroll 1
  for (int ybase = y0; ybase < y1; ybase += F3_RING) {
    // anchor rows 32 j > 0: the column sums restart from the fresh sum of the 15 window rows in row order
    // (at the top of a 16-row period slot s holds row ybase - 7 + s)
    if (!(ST_ABLATE & 8) && ybase > y0 && (ybase - y0) % F3_ANCHOR == 0) {
#pragma unroll
      for (int c = 0; c < 5; ++c) vs[c] = 0;
#pragma unroll
      for (int s = 0; s < W; ++s) {
#pragma unroll
        for (int c = 0; c < 5; ++c) vs[c] += (double)ring[s][c];
        __builtin_amdgcn_sched_barrier(0);  // one row's conversions live at a time
      }
    }
#pragma unroll
    for (int g = 0; g < F3_RING / F3_GROUP; ++g) {
      const int yg = ybase + g * F3_GROUP;
      if (yg >= y1) break;  // uniform over the workgroup
      // ---- flow rows of the PREVIOUS group go out ahead of this group's loads
      if (yg > y0) {
        if (!(w & 1)) {
          // even widths: a thread stores TWO adjacent pixels (16 bytes) of four of the eight rows -- half as many
          // vector-memory instructions as one pixel per thread and row (the texture-address unit is this kernel's
          // busiest resource, and an instruction costs it the same whatever its width)
          if (pair_x < w && pair_tid) {
#pragma unroll
            for (int r = 0; r < F3_GROUP / 2; ++r) {
              const int rr = pair_r0 + r;
              const float2 fa = Fs[rr][pair_pos], fb = Fs[rr][pair_pos + 1];
              f4v v4;
              v4.x = fa.x; v4.y = fa.y; v4.z = fb.x; v4.w = fb.y;
              // non-temporal: the next reader is another launch, tens of gigabytes later
              __builtin_nontemporal_store(v4, reinterpret_cast<f4v*>(fout + 2 * (size_t)((yg - F3_GROUP + rr) * w + pair_x)));
            }
          }
        } else if (writer) {
#pragma unroll
          for (int r = 0; r < F3_GROUP; ++r) {
            const float2 fv = Fs[r][vpos];
            f2v v2;
            v2.x = fv.x; v2.y = fv.y;
            __builtin_nontemporal_store(v2, reinterpret_cast<f2v*>(fout + 2 * (size_t)((yg - F3_GROUP + r) * w + x)));
          }
        }
      }
      // ---- phase 1: GROUP / RB batches back to back
      // phase 1 carries the loads: a wave in it goes ahead of the co-resident wave's phase-2 arithmetic, so
      // that the gathers are in flight as early as possible (measured: -3 % on the launch)
      __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int bb = 0; bb < F3_GROUP / RB; ++bb) {
        const int ybb = yg + bb * RB;
        const int q = (g * (F3_GROUP / RB) + bb) % D;  // queue slot of this batch: compile-time constant
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int t = g * F3_GROUP + bb * RB + r;  // row of the 16-row period: compile-time constant
          float m[5];
#if ST_ABLATE & 128
          {
            UmLoads G;
            um_fill_win(L[q][r], win, wslot, tid);
            if ((ST_ABLATE & 256) && r == 0) __syncthreads();
            um_gather_win(L[q][r], win, wslot, h, w, xc, d_clamp(ybb + r + M + 1, 0, h - 1), fcur[q][r], G);
            wslot = wslot == 2 ? 0 : wslot + 1;
            um_finish(G, h, w, xc, d_clamp(ybb + r + M + 1, 0, h - 1), fcur[q][r], m);
          }
#else
          um_finish(L[q][r], h, w, xc, d_clamp(ybb + r + M + 1, 0, h - 1), fcur[q][r], m);
#endif
#pragma unroll
          for (int c = 0; c < 5; ++c) {
            Vs[bb * RB + r][c][vpos] = (float)vs[c];
            const float d = (ST_ABLATE & 8) ? m[c] * 0.5f : m[c] - ring[t % F3_RING][c];  // ablation 8: no ring
            vs[c] += d;
            if (!(ST_ABLATE & 8)) ring[(t + W) % F3_RING][c] = m[c];
          }
          fcur[q][r] = fnext[r];
          if (ST_ABLATE & 4) {  // ablation: no expansion loads
            L[q][r].q = make_float4(fnext[r].x, fnext[r].y, m[0], m[1]); L[q][r].qs = m[2];
            L[q][r].t0 = L[q][r].t1 = L[q][r].b0 = L[q][r].b1 = L[q][r].q;
            L[q][r].ts.x = L[q][r].ts.y = L[q][r].bs.x = L[q][r].bs.y = m[3];
          } else {
#if ST_ABLATE & 128
            um_issue_win(R0, R1, np, h, w, xc, d_clamp(ybb + D * RB + r + M + 1, 0, h - 1), L[q][r]);
#else
            um_issue(R0, R1, np, h, w, xc, d_clamp(ybb + D * RB + r + M + 1, 0, h - 1), fcur[q][r], L[q][r]);
#endif
          }
        }
        // flows: the loads requested one batch ago become vectors now and the next rows are requested --
        // never a wait on loads issued in the same batch
        if (MODE == FLOW_COARSE2) {
          int ysb[RB];
#pragma unroll
          for (int r = 0; r < RB; ++r) ysb[r] = d_clamp(ybb + (D + 1) * RB + r + M + 1, 0, h - 1);
          coarse2_finish_batch<RB>(a, cx, ysb[0], ysb, rawn, fnext);
          coarseN_issue<RB>(a, C, cx, d_clamp(ybb + (D + 2) * RB + M + 1, 0, h - 1), rawn);
        } else {
#pragma unroll
          for (int r = 0; r < RB; ++r) fnext[r] = flow_finish<MODE>(a, fin, C, cx, d_clamp(ybb + (D + 1) * RB + r + M + 1, 0, h - 1), raw[MODE == FLOW_COARSE2 ? 0 : r]);
#pragma unroll
          for (int r = 0; r < RB; ++r) flow_issue<MODE>(a, fin, C, cx, xc, d_clamp(ybb + (D + 2) * RB + r + M + 1, 0, h - 1), raw[MODE == FLOW_COARSE2 ? 0 : r]);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      // ---- phase 2: horizontal window sums + solve; thread = (row of the group, 8-pixel segment)
      if (tid < F3_GROUP * F3_NSEG) {
        const int r = tid / F3_NSEG, sg = tid - r * F3_NSEG;
        const int j0 = B2_HALO + sg * F3_SW;          // first column of the segment (multiple of 8)
        const float* __restrict__ vrow = &Vs[r][0][j0 + (j0 >> 3)];  // f3_pos(j0 + k) = f3_pos(j0) + k + ((k + 0) >> 3) for k >= 0
        double t[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          const float* vp = vrow + c * F3_PADW;
          // columns j0 - 7 .. j0 + 7: padded offsets -8 .. -2 (columns j0-7 .. j0-1), 0 .. 7
          double acc = vp[-8];
#pragma unroll
          for (int k = -6; k <= 7; ++k) acc += (double)vp[k < 0 ? k - 1 : k];
          t[c] = acc;
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < F3_SW; ++i) {
          if (i > 0) {
            // entering column j0 + i + 7 (8 .. 14 past j0: padded +1), leaving column j0 + i - 8
#pragma unroll
            for (int c = 0; c < 5; ++c) {
              const float* vp = vrow + c * F3_PADW;
              const int ke = i + 7, kl = i - 8;  // relative to j0
              t[c] += (double)vp[ke >= 8 ? ke + 1 : ke] - (double)vp[kl < 0 ? (kl < -8 ? kl - 2 : kl - 1) : kl];
            }
          }
          const double g11 = t[0] * a.scale, g12 = t[1] * a.scale, g22 = t[2] * a.scale;
          const double h1 = t[3] * a.scale, h2 = t[4] * a.scale;
          const double idet = d_rcp_pos(g11 * g22 - g12 * g12 + 1e-3);
          Fs[r][j0 + (j0 >> 3) + i] = make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    }
  }
  // flow of the last group
  if (writer) {
    const int ylast = y0 + ((y1 - y0 - 1) / F3_GROUP) * F3_GROUP;
#pragma unroll
    for (int r = 0; r < F3_GROUP; ++r) {
      const int y = ylast + r;
      if (y < y1) *reinterpret_cast<float2*>(fout + 2 * (size_t)(y * w + x)) = Fs[r][vpos];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_flow_iter_roles: the marching iteration with its three kinds of work on three kinds of waves.
//
// Why: k_flow_iter3 needs ~230 registers (the 16-row ring of M, the running sums, a batch of gathers in flight, the
// solver's temporaries), so two of its waves share a SIMD -- and a gfx950 wave issues at most one vector-ALU instruction
// per ~8 clocks (scripts/ubench/valu1wave.hip): a launch that leaves one workgroup per compute unit (one pair at 1080p:
// 272 workgroups) runs at half the ALU rate with every memory and LDS latency exposed, ~1 us per row.  Here a strip of
// the frame is marched by ONE workgroup of 3 x NCW waves in three roles, pipelined through LDS in steps of FR_G = 4
// rows with one barrier per step:
//   makers   thread = column: flow_in -> UpdateMatrices (the gathers, ~100 instructions per pixel) -> M row -> LDS
//   summers  thread = column: the 16-slot ring of M rows and the double running column sums (anchored every 32 rows)
//            -> column sums as float -> LDS; then, for the rows of the PREVIOUS step, thread = (row, 8-pixel segment,
//            channel): the fresh 15-term horizontal sum at the segment's first column and the four slides to its
//            fifth -> both sums as double -> LDS
//   solvers  thread = (row, half segment): starts from the handed-over sum, three slides, four 2x2 solves, flow stored
//            straight to global memory
// (the horizontal chain of a segment -- fresh sum at column 8 i, seven slides -- is the longest serial piece of the
// iteration: 700 instructions in one thread; cut this way no thread runs more than ~400 per step).  No role needs more
// than 128 registers, so 12 or 15 waves fit a compute unit; they sit on the four SIMDs round-robin (one wave of each role
// per SIMD) and interleave at the full issue rate, the solves of step s overlapping the gathers of step s + 3.
// Arithmetic, operand order and association are k_flow_iter3's (the canonical association: vertical anchors at rows
// 32 j, horizontal anchors at columns 8 i; the slides of a segment are the same operations in the same order whoever
// performs them), so the two kernels -- and k_flow_iter_tile -- agree bit for bit and the choice between them is a
// scheduling matter.
// Stream of M rows of a segment [y0, y1): index k = 0 .. 15 + (y1 - y0), row(k) = clamp(y0 - 8 + k) (k = 0 is a dummy
// that aligns output row j = k - 16 with the 4-row steps).  Step s: the makers produce k = 4 s .. 4 s + 3; the summers
// turn the rows of step s - 1 into column sums and the column sums of step s - 2 into horizontal sums; the solvers
// finish the rows whose horizontal sums were formed in step s - 1.
// LDS: M rows double-buffered, column sums in three generations (written, summed horizontally, slid by the solvers),
// horizontal sums double-buffered: 129 KB (NCW = 4), 158 KB (NCW = 5).
// ---------------------------------------------------------------------------------------------
constexpr int FR_G = 4;
template <int NCW, int NMS>
struct FrGeom {
  static constexpr int COLS = 64 * NCW, OUTMAX = COLS - 2 * B2_HALO, PADW = COLS + COLS / 8;
  static constexpr int NSEGMAX = OUTMAX / F3_SW;
  static constexpr int THREADS = (2 + NMS) * COLS;  // NMS sets of maker waves (each takes every NMS-th 2-row batch)
  static_assert(FR_G * 2 * NSEGMAX <= COLS, "one solver thread per (row, half segment)");
  static_assert(THREADS <= 1024 && (NMS == 1 || NMS == 2), "workgroup size");
};

template <int NCW, int NMS, int MODE, int RB = 2>
__global__ __launch_bounds__((FrGeom<NCW, NMS>::THREADS)) void k_flow_iter_roles(IterArgs a) {
  typedef FrGeom<NCW, NMS> G;
  constexpr int M = 7, W = 15;
  static_assert(FR_G % (RB * NMS) == 0, "a step is a whole number of batches per maker set");
  __shared__ float Mb[2][FR_G][5][G::COLS];
  __shared__ float Vs[3][FR_G][5][G::PADW];
  __shared__ double Ts[2][FR_G][2 * G::NSEGMAX][5];
  const int tid = threadIdx.x;
  const int role = tid < NMS * G::COLS ? 0 : (tid < (NMS + 1) * G::COLS ? 1 : 2);  // wave-uniform
  const int h = a.h, w = a.w;
  const int np = h * w;
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  xcd_remap(bx, by, bz);
  const int pr = bz;
  const int y0 = by * a.rows_per_seg;
  const int y1 = min(h, y0 + a.rows_per_seg);
  const int SM = 4 + (y1 - y0 + FR_G - 1) / FR_G;  // maker steps 0 .. SM - 1
  const int T = SM + 3;                            // column sums 1 .. SM, horizontal sums 6 .. SM + 1, solves 7 .. SM + 2
  const int nseg = a.out_w / F3_SW;

  if (role == 0) {
    // ---------------- makers ----------------
    const int mset = NMS == 1 ? 0 : tid / G::COLS;  // this wave's maker set
    const int t = tid - mset * G::COLS;
    const int x = (int)bx * a.out_w - B2_HALO + t;
    // lanes beyond the strip's right halo (a strip may be narrower than the waves that march it) repeat its last column:
    // the same cache lines, no traffic of their own
    const int xc = d_clamp(x, 0, min(w - 1, (int)bx * a.out_w + a.out_w + B2_HALO - 1));
    const float* __restrict__ R0;
    const float* __restrict__ R1;
    if (a.pairs) {
      R0 = a.R + (size_t)a.pairs[2 * pr] * 5 * (size_t)np;
      R1 = a.R + (size_t)a.pairs[2 * pr + 1] * 5 * (size_t)np;
    } else {
      R0 = a.R;
      R1 = a.R1_direct;
    }
    const float* __restrict__ fin = a.flow_in ? a.flow_in + (size_t)pr * 2 * (size_t)np : nullptr;
    const float* __restrict__ C = a.coarse ? a.coarse + (size_t)pr * 2 * (size_t)a.ch * a.cw : nullptr;
    const CoarseX cx = (MODE == FLOW_COARSE || MODE == FLOW_COARSE2) ? coarse_x(a, xc) : CoarseX{0, 1.f, 0.f, false};
    // This set's n-th batch is rows k = KS n + RB mset + r, r < RB: with two sets a step's four rows are one batch of
    // each set (twice the gathers in flight per compute unit: the makers are what a step waits for).
    constexpr int KS = RB * NMS, NB = FR_G / KS;  // batch stride in rows, batches per step and set
    auto rowk = [&](int k) { return d_clamp(y0 - 8 + k, 0, h - 1); };
    const int kb = RB * mset;
    // software pipeline over the batches (as k_flow_iter3): gathers of the next batch in flight while this one is
    // finished, flows one batch further ahead, their loads one more
    float2 fcur[RB], fnext[RB];
    UmLoads L[RB];
    FlowRaw raw[MODE == FLOW_COARSE2 ? 1 : RB];
    FlowRawN<RB> rawn;  // FLOW_COARSE2 (coarse level exactly half as tall): the batch's rows share three coarse rows
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      fcur[r] = iter_flow_at<MODE>(a, fin, C, cx, xc, rowk(kb + r));
      um_issue(R0, R1, np, h, w, xc, rowk(kb + r), fcur[r], L[r]);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) fnext[r] = iter_flow_at<MODE>(a, fin, C, cx, xc, rowk(kb + KS + r));
    if (MODE == FLOW_COARSE2) {
      coarseN_issue<RB>(a, C, cx, rowk(kb + 2 * KS), rawn);
    } else {
#pragma unroll
      for (int r = 0; r < RB; ++r) flow_issue<MODE>(a, fin, C, cx, xc, rowk(kb + 2 * KS + r), raw[r]);
    }
    for (int sb = 0; sb < T; sb += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s = sb + u;
        if (s < T) {
          if (s < SM) {
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
              const int k0 = FR_G * s + bb * KS + kb;   // first row of this batch
              const int i0 = (NMS == 1 ? bb * RB : kb);  // its row within the step
#pragma unroll
              for (int r = 0; r < RB; ++r) {
                float m[5];
                um_finish(L[r], h, w, xc, rowk(k0 + r), fcur[r], m);
#pragma unroll
                for (int c = 0; c < 5; ++c) Mb[u][i0 + r][c][t] = m[c];
                fcur[r] = fnext[r];
                um_issue(R0, R1, np, h, w, xc, rowk(k0 + KS + r), fcur[r], L[r]);
              }
              if (MODE == FLOW_COARSE2) {
                int ysb[RB];
#pragma unroll
                for (int r = 0; r < RB; ++r) ysb[r] = rowk(k0 + 2 * KS + r);
                coarse2_finish_batch<RB>(a, cx, ysb[0], ysb, rawn, fnext);
                coarseN_issue<RB>(a, C, cx, rowk(k0 + 3 * KS), rawn);
              } else {
#pragma unroll
                for (int r = 0; r < RB; ++r) fnext[r] = flow_finish<MODE>(a, fin, C, cx, rowk(k0 + 2 * KS + r), raw[MODE == FLOW_COARSE2 ? 0 : r]);
#pragma unroll
                for (int r = 0; r < RB; ++r) flow_issue<MODE>(a, fin, C, cx, xc, rowk(k0 + 3 * KS + r), raw[MODE == FLOW_COARSE2 ? 0 : r]);
              }
            }
          }
          __syncthreads();
        }
      }
    }
  } else if (role == 1) {
    // ---------------- summers ----------------
    const int t = tid - NMS * G::COLS;
    const int vpos = f3_pos(t);
    float ring[F3_RING][5];
    double vs[5];
#pragma unroll
    for (int s2 = 0; s2 < F3_RING; ++s2)
#pragma unroll
      for (int c = 0; c < 5; ++c) ring[s2][c] = 0.f;
#pragma unroll
    for (int c = 0; c < 5; ++c) vs[c] = 0;
    // horizontal-sum items of a step: (row, segment, channel), segment fastest; a thread's items are the same in every
    // step, so their LDS offsets are formed once (the division by the run-time segment count is ~25 instructions)
    const int nitem = FR_G * nseg * 5;
    constexpr int NR = (FR_G * G::NSEGMAX * 5 + G::COLS - 1) / G::COLS;
    int it_off[NR];  // low 16 bits: float offset of f3_pos(8 + 8 sg) within a generation of Vs; high: double offset within a buffer of Ts
    static_assert(FR_G * 5 * G::PADW < 65536 && FR_G * 2 * G::NSEGMAX * 5 < 32768, "packed item offsets");
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int e = t + G::COLS * j;
      const int sg = e % nseg, rc = e / nseg, c = rc % 5, r = rc / 5;
      it_off[j] = e < nitem ? (((r * 5 + c) * G::PADW + 9 + 9 * sg) | (((r * 2 * G::NSEGMAX + 2 * sg) * 5 + c) << 16)) : -1;
    }
    int vgen = 0;  // generation (mod 3) of Vs this step's column sums go to
    for (int sb = 0; sb < T; sb += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int s = sb + u;
        if (s < T) {
          const int c4 = (u + 3) & 3;  // (s - 1) mod 4: compile-time
          // ---- (i) column sums of the rows the makers produced in step s - 1
          if (s >= 1 && s <= SM) {
            const int cstep = s - 1;
            if (cstep < 4) {
              // prologue: rows k = 4 cstep + i into ring slot k (cstep == c4 here)
#pragma unroll
              for (int i = 0; i < FR_G; ++i)
#pragma unroll
                for (int c = 0; c < 5; ++c) ring[4 * c4 + i][c] = Mb[c4 & 1][i][c][t];
              if (c4 == 3) {
                // window of output row 0 = slots 1 .. 15 (rows y0 - 7 .. y0 + 7)
                if (y0 == 0) {
                  // reference initialisation order: float(M[0]*(m+2)) + sum_{1..m-1} M[y] + float(M[m] - M[0])
#pragma unroll
                  for (int c = 0; c < 5; ++c) vs[c] = (double)(ring[M + 1][c] * (float)(M + 2));
#pragma unroll
                  for (int yy = 1; yy < M; ++yy)
#pragma unroll
                    for (int c = 0; c < 5; ++c) vs[c] += (double)ring[M + 1 + yy][c];
#pragma unroll
                  for (int c = 0; c < 5; ++c) {
                    const float d = ring[2 * M + 1][c] - ring[M + 1][c];
                    vs[c] += d;
                  }
                } else {
#pragma unroll
                  for (int c = 0; c < 5; ++c) vs[c] = 0;
#pragma unroll
                  for (int s2 = 1; s2 <= W; ++s2)
#pragma unroll
                    for (int c = 0; c < 5; ++c) vs[c] += (double)ring[s2][c];
                }
              }
            } else {
              // entering rows k = 4 cstep + i, output rows j = k - 16; ring slot of k = 4 c4 + i
              if (c4 == 0 && cstep > 4 && ((FR_G * cstep) & 31) == 16) {
                // anchor rows 32 j > 0: fresh sum of the window (slots 1 .. 15) in row order
#pragma unroll
                for (int c = 0; c < 5; ++c) vs[c] = 0;
#pragma unroll
                for (int s2 = 1; s2 <= W; ++s2) {
#pragma unroll
                  for (int c = 0; c < 5; ++c) vs[c] += (double)ring[s2][c];
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
#pragma unroll
              for (int i = 0; i < FR_G; ++i) {
                const int slot = 4 * c4 + i;
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                  const float m = Mb[c4 & 1][i][c][t];
                  Vs[vgen][i][c][vpos] = (float)vs[c];
                  const float d = m - ring[(slot + 1) & 15][c];
                  vs[c] += d;
                  ring[slot][c] = m;
                }
              }
            }
          }
          // ---- (ii) horizontal sums of the column sums written in step s - 1 (generation vgen - 1): per (row, segment,
          // channel) the fresh 15-term sum at the segment's first column, then the four slides to its fifth
          if (s >= 6 && s <= SM + 1) {
            const int pg = vgen == 0 ? 2 : vgen - 1;
            const float* __restrict__ vbase = &Vs[pg][0][0][0];
            double* __restrict__ tbase = &Ts[u & 1][0][0][0];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
              if (it_off[j] >= 0) {
                const float* __restrict__ vp = vbase + (it_off[j] & 0xffff);
                double* __restrict__ tp = tbase + (it_off[j] >> 16);
                double acc = vp[-8];
#pragma unroll
                for (int k = -6; k <= 7; ++k) acc += (double)vp[k < 0 ? k - 1 : k];
                tp[0] = acc;
#pragma unroll
                for (int i = 1; i <= 4; ++i) acc += (double)vp[8 + i] - (double)vp[i - 9];
                tp[5] = acc;
              }
            }
          }
          if (s >= 5 && s <= SM) vgen = vgen == 2 ? 0 : vgen + 1;
          __syncthreads();
        }
      }
    }
  } else {
    // ---------------- solvers ----------------
    const int t = tid - (NMS + 1) * G::COLS;
    const int nq = 2 * nseg;
    const int r = t / nq, q = t - r * nq;
    const int sg = q >> 1, hf = q & 1;
    const bool item = t < FR_G * nq;
    float* fout = a.flow_ptrs ? st_gl(a.flow_ptrs[pr]) : a.flow_out + (size_t)pr * 2 * (size_t)np;
    const int x0 = (int)bx * a.out_w + sg * F3_SW + 4 * hf;   // frame column of the first of this thread's four pixels
    const bool wide = !(w & 1) && (((uintptr_t)fout) & 15) == 0;
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    int rgen = 0;  // generation of Vs the solver reads
    for (int sb = 0; sb < T; sb += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s = sb + u;
        if (s < T) {
          const int y = y0 + FR_G * (s - 7) + r;
          if (s >= 7 && item && y < y1 && x0 < w) {
            // sums formed in step s - 1 (Ts[(s - 1) & 1]) over the column sums of generation rgen
            const float* __restrict__ vrow = &Vs[rgen][r][0][9 + 9 * sg + 4 * hf];
            double tt[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) tt[c] = Ts[(u + 1) & 1][r][q][c];
            float2 res[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (i > 0) {
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                  const float* vp = vrow + c * G::PADW;
                  tt[c] += (double)vp[8 + i] - (double)vp[i - 9];
                }
              }
              const double g11 = tt[0] * a.scale, g12 = tt[1] * a.scale, g22 = tt[2] * a.scale;
              const double h1 = tt[3] * a.scale, h2 = tt[4] * a.scale;
              const double idet = d_rcp_pos(g11 * g22 - g12 * g12 + 1e-3);
              res[i] = make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
            }
            float* dst = fout + 2 * ((size_t)y * w + x0);
            if (wide && x0 + 4 <= w) {
#pragma unroll
              for (int i = 0; i < 4; i += 2) {
                f4v v4;
                v4.x = res[i].x; v4.y = res[i].y; v4.z = res[i + 1].x; v4.w = res[i + 1].y;
                __builtin_nontemporal_store(v4, reinterpret_cast<f4v*>(dst + 2 * i));
              }
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (x0 + i < w) {
                  f2v v2;
                  v2.x = res[i].x; v2.y = res[i].y;
                  __builtin_nontemporal_store(v2, reinterpret_cast<f2v*>(dst + 2 * i));
                }
            }
          }
          if (s >= 7) rgen = rgen == 2 ? 0 : rgen + 1;
          __syncthreads();
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_flow_iter_tile: the same iteration for launches too small to fill the chip by marching
// (few pairs, coarse pyramid levels).  A workgroup produces one 32 x 32 tile: UpdateMatrices on
// the tile plus its 7-pixel apron (replicated at the frame border, as the box filter's
// BORDER_REPLICATE requires) into LDS, then the window sums in EXACTLY the association of
// k_flow_iter3 -- vertically: anchored at rows 32 j (fresh sum of the 15 rows in row order; the
// reference's initialisation order at row 0), float-rounded row differences added in double in
// between, handed over as float; horizontally: fresh 15-term sum in double at columns 8 i, seven
// slides after it -- and the same solve.  Tiles start at multiples of 32 in both directions, so
// the anchors fall on the tile's first row and on every eighth tile column, and the two kernels
// agree bit for bit: which of them a launch takes is a scheduling matter only.
// 2.1x redundant UpdateMatrices work, but thousands of short independent workgroups instead of a
// few dozen long ones: a level-3 launch drops from ~25 us to ~10 us.
// ---------------------------------------------------------------------------------------------
constexpr int FT_T = 32, FT_M = 7, FT_S = FT_T + 2 * FT_M;  // tile side, window radius, tile + apron
static_assert(FT_T == F3_ANCHOR && FT_T % F3_SW == 0 && B2_OUT % F3_SW == 0, "tile anchors must coincide with k_flow_iter3's");

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k_flow_iter_tile(IterArgs a) {
  constexpr int W = 2 * FT_M + 1;
  constexpr int NSEG = FT_T / F3_SW;  // 8-pixel segments per tile row
  __shared__ float Mt[5][FT_S][FT_S];            // M on the tile + apron; rows 0 .. 31 become the column sums in place
  __shared__ double Tt[FT_T][2 * NSEG][5];       // horizontal sums at the first and the fifth pixel of every segment
  const int tid = threadIdx.x;
  const int h = a.h, w = a.w;
  const int np = h * w;
  const int pr = blockIdx.z;
  const int X0 = blockIdx.x * FT_T, Y0 = blockIdx.y * FT_T;
  const float* __restrict__ R0;
  const float* __restrict__ R1;
  if (a.pairs) {
    R0 = a.R + (size_t)a.pairs[2 * pr] * 5 * (size_t)np;
    R1 = a.R + (size_t)a.pairs[2 * pr + 1] * 5 * (size_t)np;
  } else {
    R0 = a.R;
    R1 = a.R1_direct;
  }
  const float* __restrict__ fin = a.flow_in ? a.flow_in + (size_t)pr * 2 * (size_t)np : nullptr;
  const float* __restrict__ C = a.coarse ? a.coarse + (size_t)pr * 2 * (size_t)a.ch * a.cw : nullptr;
  float* fout = a.flow_ptrs ? st_gl(a.flow_ptrs[pr]) : a.flow_out + (size_t)pr * 2 * (size_t)np;

  // ---- phase 1: M on the tile + apron (Mt row j = source row Y0 - 7 + j, clamped).  A thread has up to NI of the
  // 46 x 46 pixels.  Their flow vectors are requested together, then the expansions in chunks of up to three pixels: a few
  // memory round trips per tile instead of two per pixel -- these launches are latency-bound, so that is most of their time.
  constexpr int NI = (FT_S * FT_S + NT - 1) / NT;
  int px[NI], py[NI];
  FlowRaw raw[NI];
  CoarseX cxs[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int i = min(tid + NT * k, FT_S * FT_S - 1);  // the last, partial round recomputes the final pixel: harmless
    const int ty = i / FT_S, tx = i - ty * FT_S;
    px[k] = d_clamp(X0 - FT_M + tx, 0, w - 1);
    py[k] = d_clamp(Y0 - FT_M + ty, 0, h - 1);
    cxs[k] = (MODE == FLOW_COARSE) ? coarse_x(a, px[k]) : CoarseX{0, 1.f, 0.f, false};
    flow_issue<MODE>(a, fin, C, cxs[k], px[k], py[k], raw[k]);
  }
  float2 fl[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) fl[k] = flow_finish<MODE>(a, fin, C, cxs[k], py[k], raw[k]);
#pragma unroll
  for (int k0 = 0; k0 < NI; k0 += 3) {
    constexpr int CH = 3;
    UmLoads L[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j)
      if (k0 + j < NI) um_issue(R0, R1, np, h, w, px[k0 + j], py[k0 + j], fl[k0 + j], L[j]);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (k0 + j < NI) {
        float m[5];
        um_finish(L[j], h, w, px[k0 + j], py[k0 + j], fl[k0 + j], m);
        const int i = tid + NT * (k0 + j);
        if (i < FT_S * FT_S) {
          const int ty = i / FT_S, tx = i - ty * FT_S;
#pragma unroll
          for (int c = 0; c < 5; ++c) Mt[c][ty][tx] = m[c];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // one chunk of gathers in flight at a time
  }
  __syncthreads();
  // ---- phase 2: column sums of the tile's 32 rows from the anchor at its first row; item = (channel, column).  The sum of
  // row j replaces M of row j (its last reader is this very step of this very thread).
  for (int i = tid; i < 5 * FT_S; i += NT) {
    const int c = i / FT_S, col = i - c * FT_S;
    // the column's 46 values first (one burst of LDS reads), then the serial chain on registers: read inside the chain, every
    // step would wait for an LDS round trip (the compiler cannot move the reads above the in-place stores)
    float mv[FT_S];
#pragma unroll
    for (int j = 0; j < FT_S; ++j) mv[j] = Mt[c][j][col];
    double vs;
    if (Y0 == 0) {
      // reference initialisation order: float(M[0]*(m+2)) + sum_{1..m-1} M[y] + float(M[m] - M[0])
      vs = (double)(mv[FT_M] * (float)(FT_M + 2));
#pragma unroll
      for (int yy = 1; yy < FT_M; ++yy) vs += (double)mv[FT_M + yy];
      const float d = mv[2 * FT_M] - mv[FT_M];
      vs += d;
    } else {
      vs = 0;
#pragma unroll
      for (int s2 = 0; s2 < W; ++s2) vs += (double)mv[s2];
    }
#pragma unroll
    for (int j = 0; j < FT_T; ++j) {
      Mt[c][j][col] = (float)vs;
      if (j + 1 < FT_T) {
        const float d = mv[j + W] - mv[j];
        vs += d;
      }
    }
  }
  __syncthreads();
  // ---- phase 3a: horizontal sums; item = (row, 8-pixel segment, channel), segments start at x = 8 i: the fresh 15-term sum
  // at the segment's first pixel and the four slides to its fifth (the longest serial chain of the tile, cut in two)
  for (int i = tid; i < FT_T * NSEG * 5; i += NT) {
    const int g = i % NSEG, rc = i / NSEG, c = rc % 5, r = rc / 5;
    const float* __restrict__ vp = &Mt[c][r][F3_SW * g];  // column j0 - 7 of the segment, j0 = FT_M + 8 g
    double acc = vp[0];
#pragma unroll
    for (int k = 1; k < W; ++k) acc += (double)vp[k];
    Tt[r][2 * g][c] = acc;
#pragma unroll
    for (int k = 1; k <= 4; ++k) acc += (double)vp[k + 2 * FT_M] - (double)vp[k - 1];
    Tt[r][2 * g + 1][c] = acc;
  }
  __syncthreads();
  // ---- phase 3b: item = (row, half segment): three more slides, four solves
  for (int i = tid; i < FT_T * 2 * NSEG; i += NT) {
    const int r = i / (2 * NSEG), q = i - r * (2 * NSEG);
    const int y = Y0 + r;
    if (y >= h) continue;
    const int j0 = FT_M + 4 * q;  // Mt column of this item's first pixel
    double t[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) t[c] = Tt[r][q][c];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k > 0) {
#pragma unroll
        for (int c = 0; c < 5; ++c) t[c] += (double)Mt[c][r][j0 + k + FT_M] - (double)Mt[c][r][j0 + k - FT_M - 1];
      }
      const double g11 = t[0] * a.scale, g12 = t[1] * a.scale, g22 = t[2] * a.scale;
      const double h1 = t[3] * a.scale, h2 = t[4] * a.scale;
      const double idet = d_rcp_pos(g11 * g22 - g12 * g12 + 1e-3);
      const int x = X0 + 4 * q + k;
      if (x < w)
        *reinterpret_cast<float2*>(fout + 2 * ((size_t)y * w + x)) =
            make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
int check_params(st_ctx* ctx, const st_fb_params& p, int h, int w) {
  if (h <= 0 || w <= 0) return st_set_error(ctx, ST_ERR_INVALID, "farneback: bad frame size %dx%d", w, h);
  if (p.flags != 0) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "farneback: flags=%d (only 0 implemented)", p.flags);
  if (p.fast_pyramids) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "farneback: fastPyramids not implemented");
  if (!(p.pyr_scale > 0 && p.pyr_scale < 1)) return st_set_error(ctx, ST_ERR_INVALID, "farneback: pyr_scale must be in (0,1)");
  if (p.poly_n != 5 && p.poly_n != 7) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "farneback: poly_n=%d (5 or 7)", p.poly_n);
  if (p.win_size < 1 || p.win_size > 63 || !(p.win_size & 1))
    return st_set_error(ctx, ST_ERR_UNSUPPORTED, "farneback: win_size=%d (odd, <= 63)", p.win_size);
  if (p.num_iters < 1 || p.num_levels < 0) return st_set_error(ctx, ST_ERR_INVALID, "farneback: bad iteration/level count");
  if (p.gray_bits != 14 && p.gray_bits != 15) return st_set_error(ctx, ST_ERR_INVALID, "farneback: gray_bits must be 14 or 15");
  if ((long long)h * w > 200000000LL) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "farneback: frames above 200 Mpx are not supported (32-bit plane offsets)");
  int levels = fb_levels(h, w, p);
  for (int k = 0; k <= levels; ++k) {
    LevelGeom g = fb_level_geom(h, w, p, k);
    if (g.ksize > kMaxTaps - 1) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "farneback: pyramid kernel size %d too large", g.ksize);
    if (g.lh < 1 || g.lw < 1) return st_set_error(ctx, ST_ERR_INVALID, "farneback: empty pyramid level");
  }
  return ST_OK;
}

int rows_per_segment(st_ctx* ctx, int h, int strips, int batch, int halo) {
  // enough workgroups to fill the chip (>= 8 per CU) without making the per-segment halo dominate
  long long target = (long long)ctx->num_cus * 8;
  long long per = (long long)strips * batch;
  long long segs = (target + per - 1) / per;
  if (segs < 1) segs = 1;
  int rows = (int)((h + segs - 1) / segs);
  int min_rows = 4 * halo;
  if (rows < min_rows) rows = min_rows;
  if (rows > h) rows = h;
  return rows;
}

// Segment height of the polynomial expansion (a scheduling choice: the vertical pass is a direct 11-tap sum, so a
// pixel's value does not depend on where its segment starts).  Large launches take whole-height segments; a launch
// of a few frames is latency-bound (a workgroup marches its rows one barrier per 4 rows), so it is cut into many
// short segments even though each re-reads 2N rows of context.
int polyexp_rows(st_ctx* ctx, int h, int strips, int n, int poly_n) {
  static const int env_min = getenv("ST_PE_MINROWS") ? atoi(getenv("ST_PE_MINROWS")) : 0;
  const int min_rows = env_min > 0 ? (env_min + PE_RB - 1) / PE_RB * PE_RB : 12;  // measured at 1-8 pairs of 1080p: 12 rows +4 % over the former 44, nothing below
  const long long target = (long long)ctx->num_cus * 8, per = (long long)strips * n;
  long long segs = (target + per - 1) / per;
  if (segs < 1) segs = 1;
  int rows = (int)((h + segs - 1) / segs);
  rows = (rows + PE_RB - 1) / PE_RB * PE_RB;
  if (rows < min_rows) rows = min_rows;
  if (rows > h) rows = h;
  return rows;
}

int launch_gray(st_ctx* ctx, const uint8_t* const* frames_table_dev, int n, int h, int w, int bits, uint8_t* gray,
                bool frames_aligned4, const TablesArg* tables = nullptr) {
  GrayArgs a;
  a.frames = frames_table_dev;
  a.gray = gray;
  a.npix = h * w;
  if (bits == 14) { a.cb = 1868; a.cg = 9617; a.cr = 4899; } else { a.cb = 3735; a.cg = 19235; a.cr = 9798; }
  a.shift = bits;
  a.rnd = 1 << (bits - 1);
  const bool vec = frames_aligned4 && a.npix % 4 == 0 && ((uintptr_t)gray & 3) == 0;
  int bx = ((vec ? a.npix / 4 : a.npix) + 255) / 256;
  if (bx > 2048) bx = 2048;
  if (bx < 1) bx = 1;
  st_timed t(ctx, ST_K_GRAY);
  if (tables && vec) hipLaunchKernelGGL(k_gray_tab<true>, dim3(bx, n), dim3(256), 0, ctx->stream, a, *tables);
  else if (tables) hipLaunchKernelGGL(k_gray_tab<false>, dim3(bx, n), dim3(256), 0, ctx->stream, a, *tables);
  else if (vec) hipLaunchKernelGGL(k_gray4, dim3(bx, n), dim3(256), 0, ctx->stream, a);
  else hipLaunchKernelGGL(k_gray, dim3(bx, n), dim3(256), 0, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

int launch_pyr(st_ctx* ctx, const uint8_t* gray, int n, int h, int w, const LevelGeom& g, float* img) {
  PyrArgs a;
  memset(&a, 0, sizeof(a));
  a.gray = gray; a.img = img;
  a.sh = h; a.sw = w; a.dh = g.lh; a.dw = g.lw; a.ks = g.ksize;
  gaussian_kernel(g.ksize, g.sigma, a.taps);
  const double inv_sx = (double)g.lw / w, inv_sy = (double)g.lh / h;
  a.scale_x = 1. / inv_sx; a.scale_y = 1. / inv_sy;
  if (g.lh == h && g.lw == w && g.ksize == 3 && h >= 2 && w >= 8 && !getenv("ST_PYR_GENERIC")) {
    Pyr0Args z;
    z.gray = gray; z.img = img; z.h = h; z.w = w; z.k0 = a.taps[1]; z.k1 = a.taps[2];
    const int bx = ((w + 3) / 4 + 255) / 256;
    long long segs = ((long long)ctx->num_cus * 8 + (long long)bx * n - 1) / ((long long)bx * n);
    int rows = (int)((h + segs - 1) / segs);
    if (rows < 32) rows = h < 32 ? h : 32;
    z.rows_per_seg = rows;
    st_timed t(ctx, ST_K_PYR);
    hipLaunchKernelGGL(k_pyr0, dim3(bx, (h + rows - 1) / rows, n), dim3(256), 0, ctx->stream, z);
    ST_HIP(ctx, hipGetLastError());
    return ST_OK;
  }
  {
    // exact power-of-two decimation with the reference's kernel sizes: dedicated marching kernel
    int S = 0;
    if (w == 2 * g.lw && h == 2 * g.lh && g.ksize == 3) S = 2;
    else if (w == 4 * g.lw && h == 4 * g.lh && g.ksize == 9) S = 4;
    else if (w == 8 * g.lw && h == 8 * g.lh && g.ksize == 19) S = 8;
    if (S && !getenv("ST_PYR_GENERIC")) {
      PyrDecArgs z;
      z.gray = gray; z.img = img; z.sh = h; z.sw = w; z.dh = g.lh; z.dw = g.lw;
      memcpy(z.taps, a.taps, sizeof(z.taps));
      const int bx = (g.lw + 255) / 256;
      long long segs = ((long long)ctx->num_cus * 8 + (long long)bx * n - 1) / ((long long)bx * n);
      int rows = (int)((g.lh + segs - 1) / segs);
      const int min_rows = S == 8 ? 4 : 8;
      if (rows < min_rows) rows = g.lh < min_rows ? g.lh : min_rows;
      z.rows_per_seg = rows;
      dim3 grid(bx, (g.lh + rows - 1) / rows, n);
      st_timed t(ctx, ST_K_PYR);
      if (S == 2) hipLaunchKernelGGL((k_pyr_dec<2, 3>), grid, dim3(256), 0, ctx->stream, z);
      else if (S == 4) hipLaunchKernelGGL((k_pyr_dec<4, 9>), grid, dim3(256), 0, ctx->stream, z);
      else hipLaunchKernelGGL((k_pyr_dec<8, 19>), grid, dim3(256), 0, ctx->stream, z);
      ST_HIP(ctx, hipGetLastError());
      return ST_OK;
    }
  }
  if (g.lh == h && g.lw == w) a.mode = PYR_COPY;
  else if (w == 2 * g.lw && h == 2 * g.lh) a.mode = PYR_AREA2;
  else a.mode = PYR_LINEAR;
  const int r = g.ksize / 2;
  a.max_rows = (int)std::ceil((PYR_OH - 1) * a.scale_y) + 3 + 2 * r + 1;
  a.max_cols = ((int)std::ceil((PYR_OW - 1) * a.scale_x) + 3 + 2 * r + 1 + 3) / 4 * 4;
  size_t lds = sizeof(float) * (size_t)a.max_rows * 2 * PYR_OW + (size_t)a.max_rows * a.max_cols;
  if (lds > 64 * 1024) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "farneback: pyramid tile needs %zu B of LDS", lds);
  dim3 grid((g.lw + PYR_OW - 1) / PYR_OW, (g.lh + PYR_OH - 1) / PYR_OH, n);
  st_timed t(ctx, ST_K_PYR);
  hipLaunchKernelGGL(k_pyr, grid, dim3(256), lds, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

// The one-pass pyramid applies to the reference's default geometry: four levels, each exactly half
// the previous one (sides multiples of 8), kernel sizes 3/3/9/19.  ST_PYR_UNFUSED=1 forces the
// per-level kernels (A/B runs).
bool pyr_fused_ok(int h, int w, const st_fb_params& p) {
  static const bool off = getenv("ST_PYR_UNFUSED") != nullptr || getenv("ST_PYR_GENERIC") != nullptr;
  if (off || fb_levels(h, w, p) != 3 || (h & 7) || (w & 7)) return false;
  static const int ks[4] = {3, 3, 9, 19};
  for (int k = 0; k <= 3; ++k) {
    const LevelGeom g = fb_level_geom(h, w, p, k);
    if (g.lh != (h >> k) || g.lw != (w >> k) || g.ksize != ks[k]) return false;
  }
  return true;
}

// imgs[k]: n x (h>>k)*(w>>k) floats
// gray != null: from the gray images; else from the RGB frames of the device table `frames` (4-byte
// aligned), the luma conversion folded into the loads
// skip0 (role-split instance only): level 0 is not produced -- the expansion reads the gray frame (launch_polyexp's gray)
int launch_pyr_fused(st_ctx* ctx, const uint8_t* gray, const uint8_t* const* frames, int n, int h, int w,
                     const st_fb_params& p, float* const imgs[4], bool skip0 = false) {
  PyrFusedArgs a;
  memset(&a, 0, sizeof(a));
  a.gray = gray; a.frames = frames;
  if (p.gray_bits == 14) { a.cb = 1868; a.cg = 9617; a.cr = 4899; } else { a.cb = 3735; a.cg = 19235; a.cr = 9798; }
  a.shift = p.gray_bits; a.rnd = 1 << (p.gray_bits - 1);
  a.img0 = imgs[0]; a.img1 = imgs[1]; a.img2 = imgs[2]; a.img3 = imgs[3];
  a.h = h; a.w = w;
  float* taps[4] = {a.k0, a.k1, a.k2, a.k3};
  for (int k = 0; k <= 3; ++k) {
    const LevelGeom g = fb_level_geom(h, w, p, k);
    float full[kMaxTaps];
    gaussian_kernel(g.ksize, g.sigma, full);
    for (int i = 0; i <= g.ksize / 2; ++i) {
      if (full[i] != full[g.ksize - 1 - i]) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "pyr: asymmetric Gaussian taps");
      taps[k][i] = full[i];
    }
  }
  // equal strips of at most 1024 columns (1920 -> 2 x 960 rather than 1024 + 896)
  const int strips = (w + 1023) / 1024;
  a.strip_w = ((w + strips - 1) / strips + 7) / 8 * 8;
  // each segment re-reads 16 rows of context, but the launch needs a few rounds of workgroups per
  // CU to balance (3 resident per CU): 8 per CU measured best at 257 frames of 1080p
  // (the role-split instance keeps two 12-wave workgroups per unit: 14 per CU measured best -- 257 frames of 1080p:
  // 1.71 / 1.43 / 1.30 / 1.23 / 1.27 ms at 1 / 2 / 4 / 7 / 9 segments)
  const bool roles_on = gray && ctx->pyr_roles != 0;
  long long segs = ((long long)ctx->num_cus * (roles_on ? 14 : 8) + (long long)strips * n - 1) / ((long long)strips * n);
  int rows = (int)((h + segs - 1) / segs);
  rows = (rows + 7) / 8 * 8;
  // (a few frames only: down to 16-row segments -- twice the rows are read, but the launch is
  // latency-bound and needs the workgroups)
  static const int env_min = getenv("ST_PYR_MINROWS") ? atoi(getenv("ST_PYR_MINROWS")) : 0;
  const int min_rows = (long long)strips * n * ((h + 63) / 64) >= 2LL * ctx->num_cus ? 64 : (env_min > 0 ? (env_min + 7) / 8 * 8 : 16);
  if (rows < min_rows) rows = h < min_rows ? h : min_rows;
  static const int force_rows = getenv("ST_PYR_SEGROWS") ? atoi(getenv("ST_PYR_SEGROWS")) : 0;  // experiments
  if (force_rows >= 8) rows = (force_rows + 7) / 8 * 8;
  a.rows_per_seg = rows;
  st_timed t(ctx, ST_K_PYR);
  // role-split instance (k_pyr_roles) unless ST_PYR_ROLES=0 (read at st_ctx_create): 257 frames 1.23 against 1.50 ms, two frames 25 against 43 us
  const int roles_env = ctx->pyr_roles;
  const bool roles = gray && roles_env != 0;
  if (skip0 && !roles) return st_set_error(ctx, ST_ERR_INVALID, "pyr: level 0 can only be left out of the role-split kernel");
  if (roles && skip0) hipLaunchKernelGGL(k_pyr_roles<false>, dim3(strips, (h + rows - 1) / rows, n), dim3(768), 0, ctx->stream, a);
  else if (roles) hipLaunchKernelGGL(k_pyr_roles<true>, dim3(strips, (h + rows - 1) / rows, n), dim3(768), 0, ctx->stream, a);
  else if (gray) hipLaunchKernelGGL(k_pyr_fused<false>, dim3(strips, (h + rows - 1) / rows, n), dim3(256), 0, ctx->stream, a);
  else hipLaunchKernelGGL(k_pyr_fused<true>, dim3(strips, (h + rows - 1) / rows, n), dim3(256), 0, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

// k_polyexp addresses a frame's expansion through 32-bit buffer offsets and rejects unwanted stores with an offset of 2^31:
// frames of at most 2^26 pixels (64 M; a 4K frame is 8.3 M) keep every sum of offsets below 2^32 and every rejected one
// above the resource's 20 np bytes
static bool polyexp_frame_fits(int h, int w) { return (unsigned long long)h * (unsigned long long)w <= (1ull << 26); }

// gray != null: level 0 of the default pyramid straight from the gray frames (k_polyexp_u8; img is not read)
int launch_polyexp(st_ctx* ctx, const float* img, int n, int h, int w, int poly_n, double poly_sigma, float* R,
                   const uint8_t* gray = nullptr) {
  if (!polyexp_frame_fits(h, w)) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "polyexp: frames above 64 M pixels are not supported");
  PolyArgs a;
  a.img = img; a.gray = gray; a.R = R; a.h = h; a.w = w;
  poly_prepare(poly_n, poly_sigma, &a.c);
  const int strips = (w + PE_OUT - 1) / PE_OUT;
  a.rows_per_seg = polyexp_rows(ctx, h, strips, n, poly_n);
  dim3 grid(strips, (h + a.rows_per_seg - 1) / a.rows_per_seg, n);
  st_timed t(ctx, ST_K_POLYEXP);
  if (gray && w < 8) return st_set_error(ctx, ST_ERR_INVALID, "polyexp: the gray-source instance needs rows of at least 8 pixels");
  if (gray && poly_n == 5) hipLaunchKernelGGL(k_polyexp_u8<5>, grid, dim3(256), 0, ctx->stream, a);
  else if (gray) hipLaunchKernelGGL(k_polyexp_u8<7>, grid, dim3(256), 0, ctx->stream, a);
  else if (poly_n == 5) hipLaunchKernelGGL(k_polyexp<5>, grid, dim3(256), 0, ctx->stream, a);
  else hipLaunchKernelGGL(k_polyexp<7>, grid, dim3(256), 0, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

// Expansions of several levels (imgs[k], geom[k], R[k] for k in ks[0..nk)) in one launch; the listed order is the
// dispatch order (largest level first, so that the small ones fill its tail).
int launch_polyexp_ml(st_ctx* ctx, float* const* imgs, const LevelGeom* geom, float* const* R, const int* ks, int nk, int n,
                      int poly_n, double poly_sigma) {
  if (nk < 1 || nk > 4) return st_set_error(ctx, ST_ERR_INVALID, "polyexp: bad level list");
  PolyArgsML a;
  memset(&a, 0, sizeof(a));
  poly_prepare(poly_n, poly_sigma, &a.c);
  a.nlv = nk;
  int blocks = 0;
  for (int i = 0; i < nk; ++i) {
    const LevelGeom& g = geom[ks[i]];
    PolyLevel& l = a.lv[i];
    if (!polyexp_frame_fits(g.lh, g.lw)) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "polyexp: frames above 64 M pixels are not supported");
    l.img = imgs[ks[i]]; l.R = R[ks[i]]; l.h = g.lh; l.w = g.lw;
    l.strips = (g.lw + PE_OUT - 1) / PE_OUT;
    l.rows_per_seg = polyexp_rows(ctx, g.lh, l.strips, n, poly_n);
    l.block0 = blocks;
    blocks += l.strips * ((g.lh + l.rows_per_seg - 1) / l.rows_per_seg);
  }
  st_timed t(ctx, ST_K_POLYEXP);
  if (poly_n == 5) hipLaunchKernelGGL(k_polyexp_ml<5>, dim3(blocks, n), dim3(256), 0, ctx->stream, a);
  else hipLaunchKernelGGL(k_polyexp_ml<7>, dim3(blocks, n), dim3(256), 0, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

int launch_update_matrices(st_ctx* ctx, UMArgs a, int n_pairs) {
  const size_t np = (size_t)a.h * a.w;
  int bx = (int)((np + 255) / 256);
  if (bx > 4096) bx = 4096;
  st_timed t(ctx, ST_K_UPDATE_MATRICES);
  hipLaunchKernelGGL(k_update_matrices, dim3(bx, n_pairs), dim3(256), 0, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

int launch_blur(st_ctx* ctx, BlurArgs a, int n_pairs) {
  if (a.m == 7) {
    const int strips = (a.w + B2_OUT - 1) / B2_OUT;
    // whole ring periods (15 rows) per segment; >= 4 workgroups per CU when the level allows.  The
    // running sums of this (unfused) path restart per segment, so the segment height must not depend
    // on how many pairs share the launch: it is derived from the level's size alone (a pair's
    // result is then the same in any batch).
    long long segs = ((long long)ctx->num_cus * 4 + (long long)strips - 1) / ((long long)strips);
    int rows = (int)((a.h + segs - 1) / segs);
    rows = (rows + 14) / 15 * 15;
    if (rows < 15) rows = 15;
    if (rows > 135 && a.h > 135) rows = 135;
    a.rows_per_seg = rows;
    dim3 grid(strips, (a.h + rows - 1) / rows, n_pairs);
    st_timed t(ctx, ST_K_BLUR_UPDATE);
    hipLaunchKernelGGL((k_blur_update_v2<7, true, 5, 2>), grid, dim3(B2_T), 0, ctx->stream, a);
    ST_HIP(ctx, hipGetLastError());
    return ST_OK;
  }
  const int strips = (a.w + (BLUR_T - 2 * a.m) - 1) / (BLUR_T - 2 * a.m);
  a.rows_per_seg = rows_per_segment(ctx, a.h, strips, 1, 2 * a.m + 1);  // independent of n_pairs, as above
  dim3 grid(strips, (a.h + a.rows_per_seg - 1) / a.rows_per_seg, n_pairs);
  st_timed t(ctx, ST_K_BLUR_UPDATE);
  hipLaunchKernelGGL(k_blur_update, grid, dim3(BLUR_T), 0, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

int launch_flow_iter(st_ctx* ctx, IterArgs a, int n_pairs) {
  // Launches whose marching form would be a handful of short segments take the tile kernel: by
  // default when the launch covers <= 600 k pixels in total (one 1080p pair: levels 1-3).  Both
  // kernels form the window sums in the same association (anchored every 32 rows / 8 columns), so
  // the choice, like the segment height below, changes the schedule and not one bit of the result:
  // a pair's flow does not depend on how many pairs share the call.  ST_ITER_TILE=0 / 1 (read at
  // st_ctx_create) force the marching / the tile kernel (A/B runs, parity tests of each kernel).
  const bool tile = ctx->tile_mode == 1 || (ctx->tile_mode != 0 && (long long)n_pairs * a.h * a.w <= ctx->tile_px);
  if (tile && (a.h + FT_T - 1) / FT_T <= 65535) {
    dim3 grid((a.w + FT_T - 1) / FT_T, (a.h + FT_T - 1) / FT_T, n_pairs);
    st_timed t(ctx, ST_K_BLUR_UPDATE);
    static const int tile_nt = getenv("ST_TILE_NT") ? atoi(getenv("ST_TILE_NT")) : 512;  // threads per tile (256: A/B runs)
    if (tile_nt == 256) {
      if (a.coarse) hipLaunchKernelGGL((k_flow_iter_tile<FLOW_COARSE, 256>), grid, dim3(256), 0, ctx->stream, a);
      else if (a.flow_in) hipLaunchKernelGGL((k_flow_iter_tile<FLOW_FIELD, 256>), grid, dim3(256), 0, ctx->stream, a);
      else hipLaunchKernelGGL((k_flow_iter_tile<FLOW_ZERO, 256>), grid, dim3(256), 0, ctx->stream, a);
    } else {
      if (a.coarse) hipLaunchKernelGGL((k_flow_iter_tile<FLOW_COARSE, 512>), grid, dim3(512), 0, ctx->stream, a);
      else if (a.flow_in) hipLaunchKernelGGL((k_flow_iter_tile<FLOW_FIELD, 512>), grid, dim3(512), 0, ctx->stream, a);
      else hipLaunchKernelGGL((k_flow_iter_tile<FLOW_ZERO, 512>), grid, dim3(512), 0, ctx->stream, a);
    }
    ST_HIP(ctx, hipGetLastError());
    return ST_OK;
  }
  const int strips = (a.w + B2_OUT - 1) / B2_OUT;
  // Segment height = whole anchor periods (32 rows).  Two workgroups are resident per CU (register and
  // LDS budget), so a launch runs in rounds of resident workgroups that each cost their rows plus
  // the 15 rows of ring initialisation: pick the segment count that minimises rounds x (rows + 15)
  // -- one round of tall segments when the batch is large (256 pairs x 8 strips = exactly four
  // rounds), more, shorter segments when that fills a partial round.
  const long long resident = (long long)ctx->num_cus * 2;
  const int periods = (a.h + F3_ANCHOR - 1) / F3_ANCHOR;
  int rows = periods * F3_ANCHOR;
  long long rounds3 = 1, wgs3 = 1;
  double best = 1e300;
  for (int segs = 1; segs <= periods; ++segs) {
    const int r = (periods + segs - 1) / segs * F3_ANCHOR;
    const long long nseg = (a.h + r - 1) / r;
    const long long wgs = (long long)strips * n_pairs * nseg;
    const long long rounds = (wgs + resident - 1) / resident;
    const double cost = (double)rounds * (r + 15);
    if (cost < best * 0.999) { best = cost; rows = r; rounds3 = rounds; wgs3 = wgs; }
  }
  // Role-split marching kernel (k_flow_iter_roles, one workgroup of 12 or 15 waves per compute unit) for launches that
  // cannot give k_flow_iter3 its two workgroups per unit: measured at 1080p (frames/s through both ops, k_flow_iter3 ->
  // roles): 1 pair per call 2 790 -> 3 100, 2: 3 690 -> 4 240, 4: 4 990 -> 5 200, 8: 6 190 -> 6 640, 16: 7 750 -> 7 690,
  // 32: 8 570 -> 8 360, 64: 9 150 -> 8 610.  ST_ITER_ROLES=1 always, 0 never (read at st_ctx_create).
  // Another kernel instance of this process has a flow call in flight on this GPU (Scanner's pipeline_instances_per_node;
  // st_flow_call_begins): the role kernel's one 12- or 15-wave workgroup per CU cannot share a unit with the other instance's
  // launches (129 / 158 KB of LDS each), k_flow_iter3's 64 KB workgroups can.  Measured, K instances x 1 / 2 pairs per call at
  // 1080p: K = 2: 4 210 -> 4 640 / 5 510 -> 5 740 frames/s, K = 4: 4 750 -> 5 180 / 5 770 -> 6 810, K = 8: 4 870 -> 5 310 /
  // 5 910 -> 6 870 (a lone instance: 3 540 -> 3 290, which is why the choice depends on it); profiles/r6_instances_modes.txt.
  const int roles_env = ctx->roles_mode == -1 && ctx->flow_concurrent ? 0 : ctx->roles_mode;
  const int roles_ncw = ctx->roles_ncw, roles_rows = ctx->roles_rows;
  // (the first iteration of a level -- coarse-flow source, expansions not yet in any cache -- is where the role kernel's single
  // maker wave per SIMD is weakest: 8 pairs, level 0: 264 us against k_flow_iter3's 226, while its field-source launches
  // win 183 : 195; it takes those launches only when k_flow_iter3 would leave more than a third of its slots empty)
  const long long roles_limit = a.coarse ? resident * 62 / 100 : resident * 95 / 100;
  if (roles_env == 1 || (roles_env != 0 && rounds3 == 1 && wgs3 < roles_limit)) {
    int best_ncw = 0, best_out = 0, best_rows = 0;
    double bestr = 1e300;
    for (int ncw = 5; ncw >= 4; --ncw) {
      if (roles_ncw && ncw != roles_ncw) continue;
      const int outmax = 64 * ncw - 2 * B2_HALO;
      const int rstrips = (a.w + outmax - 1) / outmax;
      const int out_w = ((a.w + rstrips - 1) / rstrips + 7) / 8 * 8;
      for (int segs = 1; segs <= periods; ++segs) {
        const int r = (periods + segs - 1) / segs * F3_ANCHOR;
        const long long nseg = (a.h + r - 1) / r;
        const long long wgs = (long long)rstrips * n_pairs * nseg;
        const long long rounds = (wgs + ctx->num_cus - 1) / ctx->num_cus;
        // a round costs its rows plus the 16-row prologue and the three-step pipeline tail; the 15-wave instance's
        // step is ~1.3 of the 12-wave one's when both fill the chip (it pays when it saves a round: 1920 columns are
        // 7 strips instead of 8, 238 instead of 272 workgroups for one 1080p pair).  A step's time is mostly fixed (a
        // barrier and a round of gathers), not proportional to the strip's width: narrower strips with taller segments
        // -- less re-fetch per output row, 15 strips of 128 x 17 segments of 64 rows = 255 workgroups of a 9-wave
        // instance for one 1080p pair -- measured 40 us per level-0 launch against 35.
        const double cost = (double)rounds * (r + 16 + 3 * FR_G) * (ncw == 5 ? 1.3 : 1.0);
        if (cost < bestr * 0.999) { bestr = cost; best_ncw = ncw; best_out = out_w; best_rows = r; }
      }
    }
    if (roles_rows >= F3_ANCHOR && roles_rows % F3_ANCHOR == 0) best_rows = roles_rows;
    if (best_ncw) {
      a.rows_per_seg = best_rows;
      a.out_w = best_out;
      const int rstrips = (a.w + best_out - 1) / best_out;
      dim3 grid(rstrips, (a.h + best_rows - 1) / best_rows, n_pairs);
      st_timed t(ctx, ST_K_BLUR_UPDATE);
      const int mode = a.coarse ? (a.h == 2 * a.ch ? FLOW_COARSE2 : FLOW_COARSE) : (a.flow_in ? FLOW_FIELD : FLOW_ZERO);
      // One maker set, two rows of gathers in flight per maker wave.  Measured and not kept (the template parameters remain):
      // two maker sets of the 4-column-wave instance (16 waves: 6.8 against 5.2 ms per 256-pair level-0 launch), four rows
      // in flight (163 registers; 8 pairs per call 1.37 against 1.23 ms).
      if (best_ncw == 5) {
        const dim3 blk(FrGeom<5, 1>::THREADS);
        if (mode == FLOW_COARSE2) hipLaunchKernelGGL((k_flow_iter_roles<5, 1, FLOW_COARSE2>), grid, blk, 0, ctx->stream, a);
        else if (mode == FLOW_COARSE) hipLaunchKernelGGL((k_flow_iter_roles<5, 1, FLOW_COARSE>), grid, blk, 0, ctx->stream, a);
        else if (mode == FLOW_FIELD) hipLaunchKernelGGL((k_flow_iter_roles<5, 1, FLOW_FIELD>), grid, blk, 0, ctx->stream, a);
        else hipLaunchKernelGGL((k_flow_iter_roles<5, 1, FLOW_ZERO>), grid, blk, 0, ctx->stream, a);
      } else {
        const dim3 blk(FrGeom<4, 1>::THREADS);
        if (mode == FLOW_COARSE2) hipLaunchKernelGGL((k_flow_iter_roles<4, 1, FLOW_COARSE2>), grid, blk, 0, ctx->stream, a);
        else if (mode == FLOW_COARSE) hipLaunchKernelGGL((k_flow_iter_roles<4, 1, FLOW_COARSE>), grid, blk, 0, ctx->stream, a);
        else if (mode == FLOW_FIELD) hipLaunchKernelGGL((k_flow_iter_roles<4, 1, FLOW_FIELD>), grid, blk, 0, ctx->stream, a);
        else hipLaunchKernelGGL((k_flow_iter_roles<4, 1, FLOW_ZERO>), grid, blk, 0, ctx->stream, a);
      }
      ST_HIP(ctx, hipGetLastError());
      return ST_OK;
    }
  }
  static const int force_rows = getenv("ST_ITER_ROWS") ? atoi(getenv("ST_ITER_ROWS")) : 0;  // experiments
  if (force_rows >= F3_ANCHOR && force_rows % F3_ANCHOR == 0 && force_rows < rows) rows = force_rows;
  a.rows_per_seg = rows;
  dim3 grid(strips, (a.h + rows - 1) / rows, n_pairs);
  st_timed t(ctx, ST_K_BLUR_UPDATE);
  const int mode = a.coarse ? (a.h == 2 * a.ch ? FLOW_COARSE2 : FLOW_COARSE) : (a.flow_in ? FLOW_FIELD : FLOW_ZERO);
  if (mode == FLOW_COARSE2) hipLaunchKernelGGL((k_flow_iter3<7, 2, FLOW_COARSE2, ST_EXP_D>), grid, dim3(B2_T), 0, ctx->stream, a);
  else if (mode == FLOW_COARSE) hipLaunchKernelGGL((k_flow_iter3<7, 2, FLOW_COARSE, ST_EXP_D>), grid, dim3(B2_T), 0, ctx->stream, a);
  else if (mode == FLOW_FIELD) hipLaunchKernelGGL((k_flow_iter3<7, 2, FLOW_FIELD, ST_EXP_D>), grid, dim3(B2_T), 0, ctx->stream, a);
  else hipLaunchKernelGGL((k_flow_iter3<7, 2, FLOW_ZERO, ST_EXP_D>), grid, dim3(B2_T), 0, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

// The fused iteration kernel applies to the reference's 15x15 window on frames whose pyramid
// levels are at least 2x2; it needs two flow fields per pair instead of two matrix fields.
bool fused_path(int h, int w, const st_fb_params& p) {
  if (p.win_size != 15 || getenv("ST_UNFUSED")) return false;
  const int levels = fb_levels(h, w, p);
  for (int k = 0; k <= levels; ++k) {
    LevelGeom g = fb_level_geom(h, w, p, k);
    if (g.lh < 2 || g.lw < 2) return false;
  }
  return true;
}

// Scratch needed to process `nf` distinct frames and `npairs` pairs in one pass.
size_t pass_bytes(int h, int w, const st_fb_params& p, int nf, int npairs) {
  const int levels = fb_levels(h, w, p);
  size_t np0 = (size_t)h * w;
  size_t maxCoarse = 0;
  for (int k = 0; k <= levels; ++k) {
    LevelGeom g = fb_level_geom(h, w, p, k);
    size_t npk = (size_t)g.lh * g.lw;
    if (k >= 1 && npk > maxCoarse) maxCoarse = npk;
  }
  size_t b = 0;
  b += st_align_up(np0 * nf);                        // gray
  b += st_align_up(sizeof(float) * np0 * nf);        // I (reused per level; level 0 when the pyramid is one pass)
  if (pyr_fused_ok(h, w, p))
    for (int k = 1; k <= 3; ++k) b += st_align_up(sizeof(float) * (np0 >> (2 * k)) * nf);  // I_1..I_3
  for (int k = 0; k <= levels; ++k) {
    LevelGeom g = fb_level_geom(h, w, p, k);
    b += st_align_up(sizeof(float) * 5 * (size_t)g.lh * g.lw * nf);  // R_k
  }
  b += 2 * st_align_up(sizeof(float) * (fused_path(h, w, p) ? 2 : 5) * np0 * npairs);  // flow or M ping/pong
  b += 2 * st_align_up(sizeof(float) * 2 * (maxCoarse ? maxCoarse : 1) * npairs);  // coarse flows
  b += st_align_up(sizeof(void*) * nf) + st_align_up(sizeof(int) * 2 * npairs) + st_align_up(sizeof(void*) * npairs);
  return b + 4096;
}

int farneback_pass(st_ctx* ctx, const uint8_t* const* frames, int nf, const int32_t* pairs, int npairs, int h, int w,
                   const st_fb_params& p, float* const* outs) {
  const int levels = fb_levels(h, w, p);
  const size_t np0 = (size_t)h * w;
  ST_TRY(st_ws_reserve(ctx, pass_bytes(h, w, p, nf, npairs)));
  uint8_t* gray = (uint8_t*)st_ws_alloc(ctx, np0 * nf);
  float* img = (float*)st_ws_alloc(ctx, sizeof(float) * np0 * nf);
  const bool pyr1 = pyr_fused_ok(h, w, p);
  float* imgs[4] = {img, nullptr, nullptr, nullptr};
  if (pyr1)
    for (int k = 1; k <= 3; ++k) {
      imgs[k] = (float*)st_ws_alloc(ctx, sizeof(float) * (np0 >> (2 * k)) * nf);
      if (!imgs[k]) return st_set_error(ctx, ST_ERR_OOM, "farneback: scratch plan exhausted");
    }
  std::vector<float*> R(levels + 1);
  std::vector<LevelGeom> geom(levels + 1);
  size_t maxCoarse = 1;
  for (int k = 0; k <= levels; ++k) {
    geom[k] = fb_level_geom(h, w, p, k);
    size_t npk = (size_t)geom[k].lh * geom[k].lw;
    if (k >= 1 && npk > maxCoarse) maxCoarse = npk;
    R[k] = (float*)st_ws_alloc(ctx, sizeof(float) * 5 * npk * nf);
  }
  float* M[2];
  const bool fused = fused_path(h, w, p);
  M[0] = (float*)st_ws_alloc(ctx, sizeof(float) * (fused ? 2 : 5) * np0 * npairs);
  M[1] = (float*)st_ws_alloc(ctx, sizeof(float) * (fused ? 2 : 5) * np0 * npairs);
  float* cflow[2];
  cflow[0] = (float*)st_ws_alloc(ctx, sizeof(float) * 2 * maxCoarse * npairs);
  cflow[1] = (float*)st_ws_alloc(ctx, sizeof(float) * 2 * maxCoarse * npairs);
  const uint8_t** d_frames = (const uint8_t**)st_ws_alloc(ctx, sizeof(void*) * nf);
  int* d_pairs = (int*)st_ws_alloc(ctx, sizeof(int) * 2 * npairs);
  float** d_outs = (float**)st_ws_alloc(ctx, sizeof(void*) * npairs);
  if (!gray || !img || !M[0] || !M[1] || !cflow[0] || !cflow[1] || !d_frames || !d_pairs || !d_outs || !R[levels])
    return st_set_error(ctx, ST_ERR_OOM, "farneback: scratch plan exhausted");
  TablesArg ta;
  const bool small_tables = nf <= kTabMax && npairs <= kTabMax;
  if (small_tables) {
    memset(&ta, 0, sizeof(ta));
    ta.d_frames = d_frames; ta.d_outs = d_outs; ta.d_pairs = d_pairs; ta.nf = nf; ta.npairs = npairs;
    for (int i = 0; i < nf; ++i) ta.frames[i] = frames[i];
    for (int i = 0; i < npairs; ++i) { ta.outs[i] = outs[i]; ta.pairs[2 * i] = pairs[2 * i]; ta.pairs[2 * i + 1] = pairs[2 * i + 1]; }
    static_assert(kTabMax <= 64, "one thread per table entry");
  } else {
    ST_HIP(ctx, hipMemcpyAsync(d_frames, frames, sizeof(void*) * nf, hipMemcpyHostToDevice, ctx->stream));
    ST_HIP(ctx, hipMemcpyAsync(d_pairs, pairs, sizeof(int) * 2 * npairs, hipMemcpyHostToDevice, ctx->stream));
    ST_HIP(ctx, hipMemcpyAsync(d_outs, outs, sizeof(void*) * npairs, hipMemcpyHostToDevice, ctx->stream));
  }

  // per-frame stages: each distinct frame once
  bool aligned4 = true;
  for (int i = 0; i < nf; ++i) aligned4 = aligned4 && ((uintptr_t)frames[i] & 3) == 0;
  // ST_PYR_FOLD_GRAY=1: the luma conversion rides on the loads of the one-pass pyramid (no gray pass, no
  // gray image in memory).  Bit-identical, but measured SLOWER (32.5 vs 31.6 ms per 256-pair step: 12
  // bytes per lane at a 12-byte lane stride, three loads per row instead of one, on a kernel that is
  // already VALU-bound), so the separate k_gray4 pass stays the default.
  const bool pyr_rgb = pyr1 && aligned4 && ctx->fold_gray;
  if (!pyr_rgb) {
    ST_TRY(launch_gray(ctx, d_frames, nf, h, w, p.gray_bits, gray, aligned4, small_tables ? &ta : nullptr));
  } else if (small_tables) {
    hipLaunchKernelGGL(k_set_tables, dim3(1), dim3(64), 0, ctx->stream, ta);
    ST_HIP(ctx, hipGetLastError());
  }
  // Level 0 straight from the gray frames (ST_POLY_U8, read at st_ctx_create): the role-split pyramid leaves level 0 out and the
  // level-0 expansion evaluates the 3 x 3 blur itself -- large calls only (the single multi-level launch keeps its float source)
  static const int single_max = getenv("ST_POLY_SINGLE_MAX") ? atoi(getenv("ST_POLY_SINGLE_MAX")) : 16;
  const bool single = pyr1 && levels >= 1 && levels <= 3 && npairs <= single_max;
  const bool poly_u8 = pyr1 && !pyr_rgb && !single && ctx->poly_u8 && ctx->pyr_roles != 0 && geom[0].ksize == 3 && geom[0].sigma <= 0 &&
                       geom[0].lh == h && geom[0].lw == w && w >= 8;
  if (pyr1) ST_TRY(launch_pyr_fused(ctx, pyr_rgb ? nullptr : gray, d_frames, nf, h, w, p, imgs, poly_u8));
  // Small batches: the expansions of all levels in ONE launch (level 0 first, the coarse levels fill its tail).  Measured
  // against the former arrangement -- level 0 on a second, low-priority stream beside the coarse levels' iterations --
  // at 1 / 2 / 4 / 8 pairs of 1080p per call: 292 / 444 / 731 / 1193 us per step against 308 / 458 / 744 / 1197: the
  // event record and wait of the second stream cost a launch's worth each, and the coarse iterations it overlapped with ran
  // two to four times slower beside the level-0 expansion whatever the stream priority.  (ST_POLY_SINGLE_MAX: pairs up to
  // which the single launch is used.)
  if (single) {
    int ks[4], nk = 0;
    for (int k = 0; k <= levels; ++k) ks[nk++] = k;
    ST_TRY(launch_polyexp_ml(ctx, imgs, geom.data(), R.data(), ks, nk, nf, p.poly_n, p.poly_sigma));
  } else {
    for (int k = levels; k >= 0; --k) {
      if (!pyr1) ST_TRY(launch_pyr(ctx, gray, nf, h, w, geom[k], img));
      ST_TRY(launch_polyexp(ctx, pyr1 ? imgs[k] : img, nf, geom[k].lh, geom[k].lw, p.poly_n, p.poly_sigma, R[k],
                            k == 0 && poly_u8 ? gray : nullptr));
    }
  }
  // per-pair stages, coarse to fine
  if (fused) {
    // fused iterations (k_flow_iter): M is never materialised; the M scratch doubles as the
    // two ping-pong flow fields of a level
    float* fbuf[2] = {M[0], M[1]};
    int cur = 0;  // cflow[cur] holds the previous (coarser) level's flow
    for (int k = levels; k >= 0; --k) {
      const int lh = geom[k].lh, lw = geom[k].lw;
      for (int it = 0; it < p.num_iters; ++it) {
        const bool last = it == p.num_iters - 1;
        IterArgs q;
        memset(&q, 0, sizeof(q));
        q.R = R[k]; q.pairs = d_pairs; q.h = lh; q.w = lw;
        q.scale = 1. / ((double)p.win_size * p.win_size);
        if (it == 0) {
          if (k < levels) {
            q.coarse = cflow[cur];
            q.ch = geom[k + 1].lh; q.cw = geom[k + 1].lw;
            q.scale_x = 1. / ((double)lw / q.cw);
            q.scale_y = 1. / ((double)lh / q.ch);
            q.mul = (float)(1. / p.pyr_scale);
          }
        } else {
          q.flow_in = fbuf[(it - 1) & 1];
        }
        if (last) {
          if (k == 0) q.flow_ptrs = d_outs; else q.flow_out = cflow[cur ^ 1];
        } else {
          q.flow_out = fbuf[it & 1];
        }
        ST_TRY(launch_flow_iter(ctx, q, npairs));
      }
      cur ^= 1;
    }
    return ST_OK;
  }
  // unfused path (window sizes other than 15, degenerate frames): materialised M, one
  // UpdateMatrices + numIters blur launches per level
  int cur = 0;  // cflow[cur] holds the previous (coarser) level's flow
  for (int k = levels; k >= 0; --k) {
    const int lh = geom[k].lh, lw = geom[k].lw;
    UMArgs u;
    memset(&u, 0, sizeof(u));
    u.R = R[k]; u.pairs = d_pairs; u.M = M[0]; u.h = lh; u.w = lw;
    if (k < levels) {
      u.ch = geom[k + 1].lh; u.cw = geom[k + 1].lw;
      u.coarse = cflow[cur];
      u.scale_x = 1. / ((double)lw / u.cw);
      u.scale_y = 1. / ((double)lh / u.ch);
      u.mul = (float)(1. / p.pyr_scale);
    }
    ST_TRY(launch_update_matrices(ctx, u, npairs));
    int mi = 0;
    for (int it = 0; it < p.num_iters; ++it) {
      const bool last = it == p.num_iters - 1;
      BlurArgs bl;
      memset(&bl, 0, sizeof(bl));
      bl.R = R[k]; bl.pairs = d_pairs; bl.Min = M[mi]; bl.Mout = M[mi ^ 1];
      bl.h = lh; bl.w = lw; bl.m = p.win_size / 2;
      bl.update = !last; bl.write_flow = last;
      bl.scale = 1. / ((double)p.win_size * p.win_size);
      if (last) {
        if (k == 0) bl.flow_ptrs = d_outs; else bl.flow = cflow[cur ^ 1];
      }
      ST_TRY(launch_blur(ctx, bl, npairs));
      mi ^= 1;
    }
    cur ^= 1;
  }
  return ST_OK;
}

}  // namespace

ST_EXPORT void st_fb_params_default(st_fb_params* p) {
  if (!p) return;
  p->num_levels = 3; p->pyr_scale = 0.5; p->fast_pyramids = 0; p->win_size = 15; p->num_iters = 3;
  p->poly_n = 5; p->poly_sigma = 1.2; p->flags = 0; p->gray_bits = 15;
}

ST_EXPORT int st_fb_levels(int h, int w, const st_fb_params* params) {
  if (!params || h <= 0 || w <= 0) return -1;
  return fb_levels(h, w, *params);
}

ST_EXPORT int st_fb_level_geom(int h, int w, const st_fb_params* params, int level, int* lh, int* lw, double* sigma,
                               int* ksize) {
  if (!params || h <= 0 || w <= 0 || level < 0) return ST_ERR_INVALID;
  LevelGeom g = fb_level_geom(h, w, *params, level);
  if (lh) *lh = g.lh;
  if (lw) *lw = g.lw;
  if (sigma) *sigma = g.sigma;
  if (ksize) *ksize = g.ksize;
  return ST_OK;
}

ST_EXPORT int st_farneback_pairs(st_ctx* ctx, const uint8_t* const* frames_dev, int n_frames, const int32_t* pairs,
                                 int n_pairs, int h, int w, const st_fb_params* params, float* const* flow_out_dev) {
  ST_TRY(st_enter(ctx));
  st_fb_params p;
  if (params) p = *params; else st_fb_params_default(&p);
  if (n_pairs < 0 || n_frames < 0) return st_set_error(ctx, ST_ERR_INVALID, "farneback: negative count");
  ST_TRY(check_params(ctx, p, h, w));
  if (n_pairs == 0) return ST_OK;
  if (!frames_dev || !pairs || !flow_out_dev) return st_set_error(ctx, ST_ERR_INVALID, "farneback: null table");
  ctx->flow_concurrent = st_flow_call_begins(ctx) && n_pairs <= 4;   // larger calls fill the chip by themselves
  for (int i = 0; i < n_pairs; ++i) {
    if (pairs[2 * i] < 0 || pairs[2 * i] >= n_frames || pairs[2 * i + 1] < 0 || pairs[2 * i + 1] >= n_frames)
      return st_set_error(ctx, ST_ERR_INVALID, "farneback: pair %d indexes outside [0,%d)", i, n_frames);
    if (!flow_out_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "farneback: output %d is null", i);
  }
  // Split into passes that fit the scratch limit.  A pass covers a run of pairs and the distinct
  // frames they touch (remapped to dense slots), so frames shared by consecutive pairs are
  // expanded once per pass.
  int start = 0;
  std::vector<int> slot(n_frames);
  while (start < n_pairs) {
    int count = n_pairs - start;
    std::vector<const uint8_t*> pf;
    std::vector<int32_t> pp;
    for (;;) {
      std::fill(slot.begin(), slot.end(), -1);
      pf.clear(); pp.clear();
      for (int i = start; i < start + count; ++i)
        for (int e = 0; e < 2; ++e) {
          int f = pairs[2 * i + e];
          if (slot[f] < 0) {
            if (!frames_dev[f]) return st_set_error(ctx, ST_ERR_INVALID, "farneback: frame %d is null", f);
            slot[f] = (int)pf.size();
            pf.push_back(frames_dev[f]);
          }
          pp.push_back(slot[f]);
        }
      if (pass_bytes(h, w, p, (int)pf.size(), count) <= ctx->ws_limit || count == 1) break;
      count = (count + 1) / 2;
    }
    ST_TRY(farneback_pass(ctx, pf.data(), (int)pf.size(), pp.data(), count, h, w, p, flow_out_dev + start));
    start += count;
  }
  return ST_OK;
}

// ---- stage-level entry points ----------------------------------------------------------------
ST_EXPORT int st_gray_u8(st_ctx* ctx, const uint8_t* rgb_dev, int h, int w, int gray_bits, uint8_t* gray_dev) {
  ST_TRY(st_enter(ctx));
  if (!rgb_dev || !gray_dev || h <= 0 || w <= 0 || (gray_bits != 14 && gray_bits != 15))
    return st_set_error(ctx, ST_ERR_INVALID, "gray: bad arguments");
  ST_TRY(st_ws_reserve(ctx, 4096));
  const uint8_t** t = (const uint8_t**)st_ws_alloc(ctx, sizeof(void*));
  ST_HIP(ctx, hipMemcpyAsync(t, &rgb_dev, sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
  return launch_gray(ctx, t, 1, h, w, gray_bits, gray_dev, ((uintptr_t)rgb_dev & 3) == 0);
}

ST_EXPORT int st_fb_pyr_image(st_ctx* ctx, const uint8_t* gray_dev, int h, int w, const st_fb_params* params, int level,
                              float* img_dev) {
  ST_TRY(st_enter(ctx));
  st_fb_params p;
  if (params) p = *params; else st_fb_params_default(&p);
  if (!gray_dev || !img_dev || level < 0) return st_set_error(ctx, ST_ERR_INVALID, "pyr: bad arguments");
  ST_TRY(check_params(ctx, p, h, w));
  LevelGeom g = fb_level_geom(h, w, p, level);
  if (g.ksize > kMaxTaps - 1 || g.lh < 1 || g.lw < 1) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "pyr: level %d unsupported", level);
  if (pyr_fused_ok(h, w, p) && level <= 3) {
    // the production path builds all four levels in one pass; run exactly that and hand out one
    const size_t np0 = (size_t)h * w;
    size_t bytes = 0;
    for (int k = 0; k <= 3; ++k) bytes += st_align_up(sizeof(float) * (np0 >> (2 * k)));
    ST_TRY(st_ws_reserve(ctx, bytes));
    float* imgs[4];
    for (int k = 0; k <= 3; ++k) {
      imgs[k] = k == level ? img_dev : (float*)st_ws_alloc(ctx, sizeof(float) * (np0 >> (2 * k)));
      if (!imgs[k]) return st_set_error(ctx, ST_ERR_OOM, "pyr: scratch exhausted");
    }
    return launch_pyr_fused(ctx, gray_dev, nullptr, 1, h, w, p, imgs);
  }
  return launch_pyr(ctx, gray_dev, 1, h, w, g, img_dev);
}

ST_EXPORT int st_fb_polyexp(st_ctx* ctx, const float* img_dev, int h, int w, int poly_n, double poly_sigma, float* r_dev) {
  ST_TRY(st_enter(ctx));
  if (!img_dev || !r_dev || h <= 0 || w <= 0) return st_set_error(ctx, ST_ERR_INVALID, "polyexp: bad arguments");
  if (poly_n != 5 && poly_n != 7) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "polyexp: poly_n=%d (5 or 7)", poly_n);
  return launch_polyexp(ctx, img_dev, 1, h, w, poly_n, poly_sigma, r_dev);
}

ST_EXPORT int st_fb_update_matrices(st_ctx* ctx, const float* r0_dev, const float* r1_dev, const float* flow_dev,
                                    const float* coarse_flow_dev, int ch, int cw, double pyr_scale, int h, int w,
                                    float* m_dev) {
  ST_TRY(st_enter(ctx));
  if (!r0_dev || !r1_dev || !m_dev || h <= 0 || w <= 0) return st_set_error(ctx, ST_ERR_INVALID, "update_matrices: bad arguments");
  UMArgs u;
  memset(&u, 0, sizeof(u));
  u.R = r0_dev; u.R1_direct = r1_dev; u.M = m_dev; u.h = h; u.w = w;
  if (coarse_flow_dev) {
    if (ch <= 0 || cw <= 0 || !(pyr_scale > 0)) return st_set_error(ctx, ST_ERR_INVALID, "update_matrices: bad coarse geometry");
    u.coarse = coarse_flow_dev; u.ch = ch; u.cw = cw;
    u.scale_x = 1. / ((double)w / cw);
    u.scale_y = 1. / ((double)h / ch);
    u.mul = (float)(1. / pyr_scale);
  } else {
    u.flow = flow_dev;
  }
  return launch_update_matrices(ctx, u, 1);
}

ST_EXPORT int st_fb_flow_iteration(st_ctx* ctx, const float* r0_dev, const float* r1_dev, const float* flow_in_dev,
                                    const float* coarse_flow_dev, int ch, int cw, double pyr_scale, int h, int w,
                                    int block_size, float* flow_out_dev) {
  ST_TRY(st_enter(ctx));
  if (!r0_dev || !r1_dev || !flow_out_dev || h <= 0 || w <= 0)
    return st_set_error(ctx, ST_ERR_INVALID, "flow_iteration: bad arguments");
  if (block_size != 15) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "flow_iteration: block_size=%d (15 only)", block_size);
  if (h < 2 || w < 2) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "flow_iteration: needs at least 2x2 pixels");
  if (flow_out_dev == flow_in_dev) return st_set_error(ctx, ST_ERR_INVALID, "flow_iteration: in-place not allowed");
  IterArgs q;
  memset(&q, 0, sizeof(q));
  q.R = r0_dev; q.R1_direct = r1_dev; q.h = h; q.w = w; q.flow_out = flow_out_dev;
  q.scale = 1. / ((double)block_size * block_size);
  if (coarse_flow_dev) {
    if (ch <= 0 || cw <= 0 || !(pyr_scale > 0)) return st_set_error(ctx, ST_ERR_INVALID, "flow_iteration: bad coarse geometry");
    q.coarse = coarse_flow_dev; q.ch = ch; q.cw = cw;
    q.scale_x = 1. / ((double)w / cw);
    q.scale_y = 1. / ((double)h / ch);
    q.mul = (float)(1. / pyr_scale);
  } else {
    q.flow_in = flow_in_dev;
  }
  return launch_flow_iter(ctx, q, 1);
}

ST_EXPORT int st_fb_update_flow_blur(st_ctx* ctx, const float* r0_dev, const float* r1_dev, const float* m_in_dev, int h,
                                     int w, int block_size, int update, float* flow_out_dev, float* m_out_dev) {
  ST_TRY(st_enter(ctx));
  if (!m_in_dev || !flow_out_dev || h <= 0 || w <= 0) return st_set_error(ctx, ST_ERR_INVALID, "update_flow_blur: bad arguments");
  if (block_size < 1 || block_size > 63 || !(block_size & 1)) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "update_flow_blur: block_size=%d", block_size);
  if (update && (!r0_dev || !r1_dev || !m_out_dev || m_out_dev == m_in_dev))
    return st_set_error(ctx, ST_ERR_INVALID, "update_flow_blur: update needs R0, R1 and a distinct m_out");
  BlurArgs b;
  memset(&b, 0, sizeof(b));
  b.R = r0_dev; b.R1_direct = r1_dev; b.Min = m_in_dev; b.Mout = m_out_dev; b.flow = flow_out_dev;
  b.h = h; b.w = w; b.m = block_size / 2; b.update = update ? 1 : 0; b.write_flow = 1;
  b.scale = 1. / ((double)block_size * block_size);
  return launch_blur(ctx, b, 1);
}
