// Sibling imgproc ops of the hot path.
//
// Blur replaces BlurKernel::execute
// (/root/reference/scannertools/scannertools_cpp/imgproc/blur_kernel_cpu.cpp:50-81): the
// reference's own k x k box filter over interleaved U8x3 frames -- window [-left, +right] with
// left = ceil(k/2.0) - 1, right = k/2, unsigned integer division by k^2, interior pixels only
// (the reference leaves the border of its output frame uninitialised; 0 is written here).
// The O(k^2) per-pixel loop becomes two separable integer running sums (exact: sums are integers,
// one division at the end), 3 B/px read + 3 B/px written.
//
// Resize replaces the cv::resize call of ResizeKernel::execute
// (/root/reference/scannertools/scannertools_cpp/imgproc/resize_kernel.cpp:68-73) for U8 frames:
// INTER_LINEAR (OpenCV's 11-bit fixed-point weights and its two-stage integer rounding, exact 2x2
// decimation rerouted to the INTER_AREA mean) and INTER_NEAREST; arithmetic as restated in
// oracle/oracle.c (orc_resize_u8), one thread per output pixel, weights recomputed per thread.
//
// ConvertColor replaces the cv::cvtColor call of ConvertColorKernel::execute
// (/root/reference/scannertools/scannertools_cpp/imgproc/convert_color_kernel.cpp:268-271) for the
// 8-bit codes BGR2RGB/RGB2BGR, BGR2GRAY, RGB2GRAY, GRAY2BGR/GRAY2RGB and BGR2HSV (OpenCV's
// integer tables, oracle/oracle.c orc_cvt_color_u8).
#include "st_internal.h"

namespace {

constexpr int BL_T = 256;        // threads; a thread owns 4 consecutive bytes of a frame row
constexpr int BL_ROWS = 16;      // output rows per workgroup tile
constexpr int BL_MAXK = 31;
constexpr int BL_TILEB = 4 * BL_T;  // tile width in bytes

struct BlurArgsK {
  const uint8_t* const* src;  // device table of (h, w, 3) frames
  uint8_t* const* dst;
  int h, w, k, left, right;
  int H;           // staged bytes ahead of the tile = 3*left, so a thread's window starts dword aligned in LDS
  int row_dwords;  // staged dwords per row
  int win_dwords;  // dwords covering the 4 + 3*(k-1) bytes a thread's four horizontal sums read
  unsigned magic;  // ceil(2^32 / k^2): sum / k^2 == umulhi(sum, magic) for sum < 2^18 (exact, see host side)
};

// LDS: the staged source rows of the tile, as bytes
__global__ __launch_bounds__(BL_T) void k_box_blur_u8c3(BlurArgsK a) {
  extern __shared__ unsigned smem[];
  const int t = threadIdx.x;
  const int nrows = BL_ROWS + a.k - 1;                   // staged rows of this tile
  unsigned* stage = smem;                                // [nrows][row_dwords]
  const int nb = 3 * a.w;                                // bytes per frame row
  const long long total = (long long)nb * a.h;
  const uint8_t* __restrict__ src = a.src[blockIdx.z];
  uint8_t* __restrict__ dst = a.dst[blockIdx.z];
  const int B0 = blockIdx.x * BL_TILEB;                  // first byte (within a row) of the tile
  const int Y0 = blockIdx.y * BL_ROWS;                   // first output row of the tile

  // ---- stage rows Y0-left .. Y0+BL_ROWS-1+right: LDS byte i of a row <-> row byte B0 - H + i.
  // Unaligned dword loads; bytes of a neighbouring row or outside the frame only feed border
  // outputs, which are forced to 0 below.
  for (int r = 0; r < nrows; ++r) {
    const int y = Y0 - a.left + r;
    const int yc = y < 0 ? 0 : (y >= a.h ? a.h - 1 : y);
    for (int d = t; d < a.row_dwords; d += BL_T) {
      const long long g = (long long)yc * nb + B0 - a.H + 4 * d;
      unsigned v;
      if (g >= 0 && g + 4 <= total) {
        typedef unsigned u32u __attribute__((aligned(1)));
        v = *reinterpret_cast<const u32u*>(src + g);
      } else {  // the dword straddles an end of the frame buffer: byte-wise, missing bytes read as 0
        v = 0;
        for (int j = 0; j < 4; ++j)
          if (g + j >= 0 && g + j < total) v |= (unsigned)src[g + j] << (8 * j);
      }
      stage[(size_t)r * a.row_dwords + d] = v;
    }
  }
  __syncthreads();
  // ---- horizontal sums of every staged row for this thread's 4 bytes
  const uint8_t* sb = reinterpret_cast<const uint8_t*>(stage);
  // horizontal sums of staged row r for this thread's 4 bytes: window byte j is staged byte
  // 4t + j, output byte c sums j = c + 3i, i = 0..k-1
  auto hsum = [&](int r, unsigned s[4]) {
    const uint8_t* row = sb + (size_t)r * a.row_dwords * 4 + 4 * t;
    s[0] = s[1] = s[2] = s[3] = 0;
    for (int i = 0; i < a.k; ++i) {
      s[0] += row[3 * i]; s[1] += row[3 * i + 1]; s[2] += row[3 * i + 2]; s[3] += row[3 * i + 3];
    }
  };
  // ---- vertical running sums, divide, store.  The row that leaves the window is summed again
  // from the staged bytes rather than kept: only the source rows live in LDS (18 KB for k = 3),
  // so eight workgroups share a CU and their load phases overlap each other's arithmetic.
  const int b = B0 + 4 * t;  // first byte of this thread within the row
  if (b >= nb) return;
  const int lo = 3 * a.left, hi = 3 * (a.w - a.right);  // interior bytes of a row: [lo, hi)
  unsigned v[4] = {0, 0, 0, 0};
  for (int r = 0; r < nrows; ++r) {
    unsigned hn[4];
    hsum(r, hn);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += hn[j];
    const int ro = r - (a.k - 1);  // output row of the tile completed by staged row r
    if (ro < 0) continue;
    const int y = Y0 + ro;
    if (y >= a.h) break;
    const bool yin = y >= a.left && y < a.h - a.right;
    uint8_t* out = dst + (size_t)y * nb + b;
    unsigned packed = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (yin && b + j >= lo && b + j < hi) packed |= (a.magic ? __umulhi(v[j], a.magic) : v[j]) << (8 * j);
    if (b + 3 < nb) {
      typedef unsigned u32u __attribute__((aligned(1)));  // rows start at any byte when 3*w % 4 != 0
      *reinterpret_cast<u32u*>(out) = packed;
    } else {
      for (int j = 0; b + j < nb; ++j) out[j] = (uint8_t)(packed >> (8 * j));
    }
    unsigned ho[4];
    hsum(ro, ho);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] -= ho[j];
  }
}

// ---- Resize -------------------------------------------------------------------------------------
enum { RS_NEAREST = 0, RS_LINEAR = 1, RS_AREA2 = 2, RS_COPY = 3 };

struct ResizeArgsK {
  const uint8_t* const* src;
  uint8_t* const* dst;
  int sh, sw, dh, dw, cn, mode;
  double scale_x, scale_y;  // source / destination size ratios as cv::resize computes them
};

// saturate_cast<short>(float): cvRound = round half to even, then saturation
__device__ __forceinline__ int rs_coef(float v) {
  const float r = rintf(v);
  return r < -32768.f ? -32768 : (r > 32767.f ? 32767 : (int)r);
}

__global__ __launch_bounds__(256) void k_resize_u8(ResizeArgsK a) {
  const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
  if (dx >= a.dw) return;
  const uint8_t* __restrict__ src = a.src[blockIdx.z];
  uint8_t* __restrict__ D = a.dst[blockIdx.z] + ((size_t)dy * a.dw + dx) * a.cn;
  const int cn = a.cn;
  const size_t srow = (size_t)a.sw * cn;
  if (a.mode == RS_COPY) {
    const uint8_t* S = src + (size_t)dy * srow + (size_t)dx * cn;
    for (int c = 0; c < cn; ++c) D[c] = S[c];
  } else if (a.mode == RS_NEAREST) {
    int sx = (int)floor(dx * a.scale_x), sy = (int)floor(dy * a.scale_y);
    sx = sx < a.sw - 1 ? sx : a.sw - 1;
    sy = sy < a.sh - 1 ? sy : a.sh - 1;
    const uint8_t* S = src + (size_t)sy * srow + (size_t)sx * cn;
    for (int c = 0; c < cn; ++c) D[c] = S[c];
  } else if (a.mode == RS_AREA2) {
    const uint8_t* S0 = src + (size_t)(2 * dy) * srow + (size_t)(2 * dx) * cn;
    const uint8_t* S1 = S0 + srow;
    for (int c = 0; c < cn; ++c) D[c] = (uint8_t)((S0[c] + S0[cn + c] + S1[c] + S1[cn + c] + 2) >> 2);
  } else {
    float fx = (float)((dx + 0.5) * a.scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= a.sw - 1) { fx = 0; sx = a.sw - 1; }
    const int a0 = rs_coef((1.f - fx) * 2048), a1 = rs_coef(fx * 2048);
    float fy = (float)((dy + 0.5) * a.scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    const int b0 = rs_coef((1.f - fy) * 2048), b1 = rs_coef(fy * 2048);
    const int y0 = sy < 0 ? 0 : (sy > a.sh - 1 ? a.sh - 1 : sy);
    const int y1 = sy + 1 < 0 ? 0 : (sy + 1 > a.sh - 1 ? a.sh - 1 : sy + 1);
    const uint8_t* S0 = src + (size_t)y0 * srow + (size_t)sx * cn;
    const uint8_t* S1 = src + (size_t)y1 * srow + (size_t)sx * cn;
    const bool two = sx + 1 < a.sw;  // the right edge takes a single tap * 2048
    for (int c = 0; c < cn; ++c) {
      const int r0 = two ? S0[c] * a0 + S0[cn + c] * a1 : S0[c] * 2048;
      const int r1 = two ? S1[c] * a0 + S1[cn + c] * a1 : S1[c] * 2048;
      D[c] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
    }
  }
}

// ---- ConvertColor ---------------------------------------------------------------------------------
struct CvtArgsK {
  const uint8_t* const* src;
  uint8_t* const* dst;
  long long npix;
  int code;                // cv::ColorConversionCodes value
  int cb, cg, cr, rnd, shift, bi;  // gray weights (bi = byte holding blue)
};

__global__ __launch_bounds__(256) void k_cvt_color_u8(CvtArgsK a) {
  __shared__ int sdiv[256], hdiv[256];
  const int t = threadIdx.x;
  if (a.code == ST_COLOR_BGR2HSV) {
    // RGB2HSV_b tables: saturate_cast<int>((255 << 12)/(1.*i)), saturate_cast<int>((180 << 12)/(6.*i))
    sdiv[t] = t ? (int)rint((255 << 12) / (1. * t)) : 0;
    hdiv[t] = t ? (int)rint((180 << 12) / (6. * t)) : 0;
    __syncthreads();
  }
  const uint8_t* __restrict__ src = a.src[blockIdx.y];
  uint8_t* __restrict__ dst = a.dst[blockIdx.y];
  for (long long i = (long long)blockIdx.x * 256 + t; i < a.npix; i += (long long)gridDim.x * 256) {
    if (a.code == ST_COLOR_BGR2RGB) {
      const uint8_t c0 = src[3 * i], c1 = src[3 * i + 1], c2 = src[3 * i + 2];
      dst[3 * i] = c2; dst[3 * i + 1] = c1; dst[3 * i + 2] = c0;
    } else if (a.code == ST_COLOR_BGR2GRAY || a.code == ST_COLOR_RGB2GRAY) {
      const int b = src[3 * i + a.bi], g = src[3 * i + 1], r = src[3 * i + (a.bi ^ 2)];
      dst[i] = (uint8_t)((b * a.cb + g * a.cg + r * a.cr + a.rnd) >> a.shift);
    } else if (a.code == ST_COLOR_GRAY2BGR) {
      const uint8_t v = src[i];
      dst[3 * i] = v; dst[3 * i + 1] = v; dst[3 * i + 2] = v;
    } else {  // BGR2HSV, hue range 180
      const int b = src[3 * i], g = src[3 * i + 1], r = src[3 * i + 2];
      const int v = max(b, max(g, r)), vmin = min(b, min(g, r));
      const int diff = v - vmin;
      const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
      const int sv = (diff * sdiv[v] + (1 << 11)) >> 12;
      int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
      hh = (hh * hdiv[diff] + (1 << 11)) >> 12;
      hh += hh < 0 ? 180 : 0;
      dst[3 * i] = (uint8_t)(hh < 0 ? 0 : (hh > 255 ? 255 : hh));
      dst[3 * i + 1] = (uint8_t)sv;
      dst[3 * i + 2] = (uint8_t)v;
    }
  }
}

}  // namespace

ST_EXPORT int st_box_blur_u8c3_batch(st_ctx* ctx, const uint8_t* const* frames_dev, int n, int h, int w, int kernel_size,
                                     uint8_t* const* out_dev) {
  ST_TRY(st_enter(ctx));
  if (n < 0 || h <= 0 || w <= 0 || (long long)h * w > 200000000LL)
    return st_set_error(ctx, ST_ERR_INVALID, "blur: bad arguments (n=%d h=%d w=%d)", n, h, w);
  if (kernel_size < 1 || kernel_size > BL_MAXK)
    return st_set_error(ctx, ST_ERR_UNSUPPORTED, "blur: kernel_size %d outside [1, %d]", kernel_size, BL_MAXK);
  if (n == 0) return ST_OK;
  if (!frames_dev || !out_dev) return st_set_error(ctx, ST_ERR_INVALID, "blur: null argument");
  for (int i = 0; i < n; ++i) {
    if (!frames_dev[i] || !out_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "blur: row %d is null", i);
    if (frames_dev[i] == out_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "blur: row %d aliases its output", i);
  }
  const size_t tb = st_align_up(sizeof(void*) * (size_t)n);
  ST_TRY(st_ws_reserve(ctx, 2 * tb));
  const uint8_t** d_src = (const uint8_t**)st_ws_alloc(ctx, tb);
  uint8_t** d_dst = (uint8_t**)st_ws_alloc(ctx, tb);
  ST_HIP(ctx, hipMemcpyAsync(d_src, frames_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  ST_HIP(ctx, hipMemcpyAsync(d_dst, out_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  BlurArgsK a;
  a.h = h; a.w = w; a.k = kernel_size;
  a.left = (kernel_size + 1) / 2 - 1;  // ceil(k/2.0) - 1
  a.right = kernel_size / 2;
  a.H = 3 * a.left;
  a.win_dwords = (4 + 3 * (kernel_size - 1) + 3) / 4;
  a.row_dwords = BL_T + a.win_dwords;  // every thread reads win_dwords dwords from dword t on
  // k^2 <= 961 and sums <= 961 * 255 < 2^18: with magic = floor(2^32 / d) + 1 the error term
  // sum * (magic * d - 2^32) <= 2^18 * 961 < 2^32, so umulhi(sum, magic) == sum / d exactly
  const unsigned div = (unsigned)(kernel_size * kernel_size);
  a.magic = div == 1 ? 0u : (unsigned)((1ull << 32) / div) + 1u;
  const int nrows = BL_ROWS + kernel_size - 1;
  const size_t lds = (size_t)nrows * a.row_dwords * 4;
  if (lds > 160 * 1024) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "blur: kernel_size %d needs %zu B of LDS", kernel_size, lds);
  ST_HIP(ctx, hipFuncSetAttribute((const void*)k_box_blur_u8c3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nb = 3 * w;
  for (int f0 = 0; f0 < n; f0 += 65535) {
    const int nf = n - f0 < 65535 ? n - f0 : 65535;
    a.src = d_src + f0; a.dst = d_dst + f0;
    dim3 grid((nb + BL_TILEB - 1) / BL_TILEB, (h + BL_ROWS - 1) / BL_ROWS, nf);
    if (grid.y > 65535) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "blur: frame too tall");
    st_timed t(ctx, ST_K_BLUR_OP);
    hipLaunchKernelGGL(k_box_blur_u8c3, grid, dim3(BL_T), lds, ctx->stream, a);
    ST_HIP(ctx, hipGetLastError());
  }
  return ST_OK;
}

ST_EXPORT int st_resize_u8_batch(st_ctx* ctx, const uint8_t* const* frames_dev, int n, int h, int w, int channels,
                                 int out_h, int out_w, int interpolation, uint8_t* const* out_dev) {
  ST_TRY(st_enter(ctx));
  if (n < 0 || h <= 0 || w <= 0 || out_h <= 0 || out_w <= 0 || channels < 1 || channels > 4 ||
      (long long)h * w > 200000000LL || (long long)out_h * out_w > 200000000LL)
    return st_set_error(ctx, ST_ERR_INVALID, "resize: bad arguments (n=%d %dx%dx%d -> %dx%d)", n, h, w, channels, out_h, out_w);
  if (interpolation != ST_INTER_NEAREST && interpolation != ST_INTER_LINEAR)
    return st_set_error(ctx, ST_ERR_UNSUPPORTED, "resize: interpolation %d (INTER_NEAREST = 0 and INTER_LINEAR = 1 are implemented)", interpolation);
  if (out_h > 65535) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "resize: output taller than 65535 rows");
  if (n == 0) return ST_OK;
  if (!frames_dev || !out_dev) return st_set_error(ctx, ST_ERR_INVALID, "resize: null argument");
  for (int i = 0; i < n; ++i)
    if (!frames_dev[i] || !out_dev[i] || frames_dev[i] == out_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "resize: row %d is null or aliased", i);
  const size_t tb = st_align_up(sizeof(void*) * (size_t)n);
  ST_TRY(st_ws_reserve(ctx, 2 * tb));
  const uint8_t** d_src = (const uint8_t**)st_ws_alloc(ctx, tb);
  uint8_t** d_dst = (uint8_t**)st_ws_alloc(ctx, tb);
  ST_HIP(ctx, hipMemcpyAsync(d_src, frames_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  ST_HIP(ctx, hipMemcpyAsync(d_dst, out_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  ResizeArgsK a;
  a.sh = h; a.sw = w; a.dh = out_h; a.dw = out_w; a.cn = channels;
  const double inv_sx = (double)out_w / w, inv_sy = (double)out_h / h;
  a.scale_x = 1. / inv_sx; a.scale_y = 1. / inv_sy;
  if (h == out_h && w == out_w) a.mode = RS_COPY;
  else if (interpolation == ST_INTER_NEAREST) a.mode = RS_NEAREST;
  else if (w == 2 * out_w && h == 2 * out_h) a.mode = RS_AREA2;
  else a.mode = RS_LINEAR;
  for (int f0 = 0; f0 < n; f0 += 65535) {
    const int nf = n - f0 < 65535 ? n - f0 : 65535;
    a.src = d_src + f0; a.dst = d_dst + f0;
    st_timed t(ctx, ST_K_RESIZE);
    hipLaunchKernelGGL(k_resize_u8, dim3((out_w + 255) / 256, out_h, nf), dim3(256), 0, ctx->stream, a);
    ST_HIP(ctx, hipGetLastError());
  }
  return ST_OK;
}

ST_EXPORT int st_cvt_color_out_channels(int code, int in_channels) {
  switch (code) {
    case ST_COLOR_BGR2RGB: case ST_COLOR_BGR2HSV: return in_channels == 3 ? 3 : -1;
    case ST_COLOR_BGR2GRAY: case ST_COLOR_RGB2GRAY: return in_channels == 3 ? 1 : -1;
    case ST_COLOR_GRAY2BGR: return in_channels == 1 ? 3 : -1;
    default: return -1;
  }
}

ST_EXPORT int st_cvt_color_u8_batch(st_ctx* ctx, const uint8_t* const* frames_dev, int n, int h, int w, int channels,
                                    int code, int gray_bits, uint8_t* const* out_dev) {
  ST_TRY(st_enter(ctx));
  if (n < 0 || h <= 0 || w <= 0 || (long long)h * w > 200000000LL)
    return st_set_error(ctx, ST_ERR_INVALID, "cvt_color: bad arguments (n=%d h=%d w=%d)", n, h, w);
  if (st_cvt_color_out_channels(code, channels) < 0)
    return st_set_error(ctx, ST_ERR_UNSUPPORTED, "cvt_color: conversion code %d on %d-channel frames is not implemented", code, channels);
  if (gray_bits != 14 && gray_bits != 15) return st_set_error(ctx, ST_ERR_INVALID, "cvt_color: gray_bits must be 14 or 15");
  if (n == 0) return ST_OK;
  if (!frames_dev || !out_dev) return st_set_error(ctx, ST_ERR_INVALID, "cvt_color: null argument");
  for (int i = 0; i < n; ++i)
    if (!frames_dev[i] || !out_dev[i] || frames_dev[i] == out_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "cvt_color: row %d is null or aliased", i);
  const size_t tb = st_align_up(sizeof(void*) * (size_t)n);
  ST_TRY(st_ws_reserve(ctx, 2 * tb));
  const uint8_t** d_src = (const uint8_t**)st_ws_alloc(ctx, tb);
  uint8_t** d_dst = (uint8_t**)st_ws_alloc(ctx, tb);
  ST_HIP(ctx, hipMemcpyAsync(d_src, frames_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  ST_HIP(ctx, hipMemcpyAsync(d_dst, out_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  CvtArgsK a;
  a.npix = (long long)h * w; a.code = code;
  if (gray_bits == 14) { a.cb = 1868; a.cg = 9617; a.cr = 4899; } else { a.cb = 3735; a.cg = 19235; a.cr = 9798; }
  a.shift = gray_bits; a.rnd = 1 << (gray_bits - 1);
  a.bi = code == ST_COLOR_RGB2GRAY ? 2 : 0;
  long long bx = (a.npix + 255) / 256;
  if (bx > 4096) bx = 4096;
  for (int f0 = 0; f0 < n; f0 += 65535) {
    const int nf = n - f0 < 65535 ? n - f0 : 65535;
    a.src = d_src + f0; a.dst = d_dst + f0;
    st_timed t(ctx, ST_K_CVT_COLOR);
    hipLaunchKernelGGL(k_cvt_color_u8, dim3((unsigned)bx, nf), dim3(256), 0, ctx->stream, a);
    ST_HIP(ctx, hipGetLastError());
  }
  return ST_OK;
}
