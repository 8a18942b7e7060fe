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
// decimation rerouted to the INTER_AREA mean), INTER_NEAREST, INTER_CUBIC (4x4 fixed-point taps) and
// INTER_AREA (integer cells, fractional cells in float, cell-aligned bilinear weights when enlarging); arithmetic as restated in
// oracle/oracle.c (orc_resize_u8), one thread per output pixel, weights recomputed per thread.
//
// ConvertColor replaces the cv::cvtColor call of ConvertColorKernel::execute
// (/root/reference/scannertools/scannertools_cpp/imgproc/convert_color_kernel.cpp:268-271) for the
// 8-bit codes BGR2RGB/RGB2BGR, BGR2GRAY, RGB2GRAY, GRAY2BGR/GRAY2RGB, BGR/RGB <-> YCrCb and BGR2HSV
// (OpenCV's integer tables, oracle/oracle.c orc_cvt_color_u8).
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

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
// KS > 0: kernel size known at compile time (the window bytes are then read as whole dwords and
// unpacked with static shifts); KS == 0: any size, byte reads.
template <int KS>
__global__ __launch_bounds__(BL_T) void k_box_blur_u8c3(BlurArgsK a) {
  extern __shared__ unsigned smem[];
  const int t = threadIdx.x;
  const int nrows = BL_ROWS + a.k - 1;                   // staged rows of this tile
  unsigned* stage = smem;                                // [nrows][row_dwords]
  const int nb = 3 * a.w;                                // bytes per frame row
  const long long total = (long long)nb * a.h;
  const uint8_t* __restrict__ src = st_gl(a.src[blockIdx.z]);
  uint8_t* __restrict__ dst = st_gl(a.dst[blockIdx.z]);
  const int B0 = blockIdx.x * BL_TILEB;                  // first byte (within a row) of the tile
  const int Y0 = blockIdx.y * BL_ROWS;                   // first output row of the tile

  // ---- stage rows Y0-left .. Y0+BL_ROWS-1+right: LDS byte i of a row <-> row byte B0 - H + i.
  // Unaligned dword loads; bytes of a neighbouring row or outside the frame only feed border
  // outputs, which are forced to 0 below.
  auto fetch = [&](int r, int d) -> unsigned {
    const int y = Y0 - a.left + r;
    const int yc = y < 0 ? 0 : (y >= a.h ? a.h - 1 : y);
    const long long g = (long long)yc * nb + B0 - a.H + 4 * d;
    if (g >= 0 && g + 4 <= total) {
      typedef unsigned u32u __attribute__((aligned(1)));
      return *reinterpret_cast<const u32u*>(src + g);
    }
    unsigned v = 0;  // the dword straddles an end of the frame buffer: byte-wise, missing bytes read as 0
    for (int j = 0; j < 4; ++j)
      if (g + j >= 0 && g + j < total) v |= (unsigned)src[g + j] << (8 * j);
    return v;
  };
  if (KS > 0) {
    // compile-time size: ALL the tile's rows are requested before any is stored to LDS, so the workgroup pays the memory
    // latency once instead of once per row (the serial form bounded the kernel: 18 dependent round trips per tile)
    constexpr int NR = BL_ROWS + (KS > 0 ? KS : 1) - 1;
    unsigned va[NR], vb[NR];
    const bool second = t + BL_T < a.row_dwords;  // the few dwords of a staged row beyond the first 256
#pragma unroll
    for (int r = 0; r < NR; ++r) va[r] = fetch(r, t);
    if (second) {
#pragma unroll
      for (int r = 0; r < NR; ++r) vb[r] = fetch(r, t + BL_T);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) stage[(size_t)r * a.row_dwords + t] = va[r];
    if (second) {
#pragma unroll
      for (int r = 0; r < NR; ++r) stage[(size_t)r * a.row_dwords + t + BL_T] = vb[r];
    }
  } else {
    // any size: the same in chunks of 8 rows
    for (int r0 = 0; r0 < nrows; r0 += 8)
      for (int d = t; d < a.row_dwords; d += BL_T) {
        unsigned v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = r0 + j < nrows ? fetch(r0 + j, d) : 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (r0 + j < nrows) stage[(size_t)(r0 + j) * a.row_dwords + d] = v8[j];
      }
  }
  __syncthreads();
  // ---- horizontal sums of every staged row for this thread's 4 bytes
  const uint8_t* sb = reinterpret_cast<const uint8_t*>(stage);
  // horizontal sums of staged row r for this thread's 4 bytes: window byte j is staged byte
  // 4t + j, output byte c sums j = c + 3i, i = 0..k-1
  auto hsum = [&](int r, unsigned s[4]) {
    s[0] = s[1] = s[2] = s[3] = 0;
    if (KS > 0) {
      // the window (3*KS + 1 bytes) starts dword aligned at staged dword t of the row
      constexpr int NW = (3 * KS + 1 + 3) / 4;
      const unsigned* rw = stage + (size_t)r * a.row_dwords + t;
      unsigned wd[NW];
#pragma unroll
      for (int q = 0; q < NW; ++q) wd[q] = rw[q];
#pragma unroll
      for (int i = 0; i < KS; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int j = 3 * i + c;
          s[c] += (wd[j >> 2] >> (8 * (j & 3))) & 0xffu;
        }
    } else {
      // any size: window byte j (0 <= j < 3k) feeds output byte j % 3, and output byte 3 is output byte 0 shifted by one tap,
      // so three residue-class sums do -- taken from whole dwords, twelve window bytes (four taps) per round with the
      // residues known at compile time, the last k % 4 taps bytewise
      const unsigned* rw = stage + (size_t)r * a.row_dwords + t;
      const uint8_t* row = sb + (size_t)r * a.row_dwords * 4 + 4 * t;
      unsigned A0 = 0, A1 = 0, A2 = 0;
      const int rounds = a.k >> 2;
      for (int m = 0; m < rounds; ++m) {
        const unsigned d0 = rw[3 * m], d1 = rw[3 * m + 1], d2 = rw[3 * m + 2];
        A0 += (d0 & 0xffu) + (d0 >> 24) + ((d1 >> 16) & 0xffu) + ((d2 >> 8) & 0xffu);
        A1 += ((d0 >> 8) & 0xffu) + (d1 & 0xffu) + (d1 >> 24) + ((d2 >> 16) & 0xffu);
        A2 += ((d0 >> 16) & 0xffu) + ((d1 >> 8) & 0xffu) + (d2 & 0xffu) + (d2 >> 24);
      }
      for (int i = 4 * rounds; i < a.k; ++i) { A0 += row[3 * i]; A1 += row[3 * i + 1]; A2 += row[3 * i + 2]; }
      s[0] = A0; s[1] = A1; s[2] = A2; s[3] = A0 - row[0] + row[3 * a.k];
    }
  };
  // ---- vertical running sums, divide, store.  The row that leaves the window is summed again
  // from the staged bytes rather than kept: only the source rows live in LDS (18 KB for k = 3),
  // so eight workgroups share a CU and their load phases overlap each other's arithmetic.
  const int b = B0 + 4 * t;  // first byte of this thread within the row
  if (b >= nb) return;
  const int lo = 3 * a.left, hi = 3 * (a.w - a.right);  // interior bytes of a row: [lo, hi)
  unsigned v[4] = {0, 0, 0, 0};
  auto emit = [&](int ro) {  // output row `ro` of the tile from the window sums in v; false: past the frame
    const int y = Y0 + ro;
    if (y >= a.h) return false;
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
    return true;
  };
  if (KS > 0) {
    // compile-time kernel size: the KS horizontal sums inside the window stay in a register ring (the loop is fully
    // unrolled, so the ring is indexed statically) -- each staged row is read and summed ONCE
    unsigned ring[KS > 0 ? KS : 1][4];
#pragma unroll
    for (int r = 0; r < BL_ROWS + KS - 1; ++r) {
      unsigned hn[4];
      hsum(r, hn);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += hn[j];
      const int ro = r - (KS - 1);
      if (ro >= 0) {
        if (!emit(ro)) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] -= ring[ro % KS][j];  // the row that leaves: staged row ro
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) ring[r % KS][j] = hn[j];    // slot of row r - KS, which left above (or was never used)
    }
    return;
  }
  for (int r = 0; r < nrows; ++r) {
    unsigned hn[4];
    hsum(r, hn);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += hn[j];
    const int ro = r - (a.k - 1);  // output row of the tile completed by staged row r
    if (ro < 0) continue;
    if (!emit(ro)) break;
    // generic sizes: the row that leaves the window is summed again from the staged bytes rather than kept
    unsigned ho[4];
    hsum(ro, ho);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] -= ho[j];
  }
}

// ---- Resize -------------------------------------------------------------------------------------
enum { RS_NEAREST = 0, RS_LINEAR = 1, RS_AREA2 = 2, RS_COPY = 3, RS_CUBIC = 4, RS_AREA_INT = 5, RS_AREA = 6, RS_LINEAR_AREA = 7, RS_LANCZOS4 = 8 };

struct ResizeArgsK {
  const uint8_t* const* src;
  uint8_t* const* dst;
  int sh, sw, dh, dw, cn, mode;
  double scale_x, scale_y;  // source / destination size ratios as cv::resize computes them
  double inv_scale_x, inv_scale_y;
  int iscale_x, iscale_y;   // RS_AREA_INT: integer cell size
  // RS_LANCZOS4: per-axis tables built on the host (cv::interpolateLanczos4 evaluates sin / cos in
  // double; the host's libm is the one OpenCV and the oracle use): first-tap offsets + 3 and 8
  // fixed-point weights per destination column / row
  const int* xofs; const short* ialpha;
  const int* yofs; const short* ibeta;
};

// saturate_cast<short>(float): cvRound = round half to even, then saturation
__device__ __forceinline__ int rs_coef(float v) {
  const float r = rintf(v);
  return r < -32768.f ? -32768 : (r > 32767.f ? 32767 : (int)r);
}

// cv::interpolateCubic, A = -0.75
__device__ __forceinline__ void rs_cubic(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

__device__ __forceinline__ uint8_t rs_sat_float(float v) {  // saturate_cast<uchar>(float): cvRound
  const float r = rintf(v);
  return (uint8_t)(r < 0.f ? 0 : (r > 255.f ? 255 : (int)r));
}

// One destination cell of computeResizeAreaTab: the source cells [first, first + n) and the weights
// of the (optional) fractional head, the full cells and the (optional) fractional tail.
struct AreaCell {
  int head_si, sx1, sx2;  // head_si < 0: no head; full cells sx1 .. sx2-1; tail at sx2 when tail_w > 0
  float head_w, full_w, tail_w;
  bool tail;
};
__device__ __forceinline__ AreaCell rs_area_cell(int dx, int ssize, double scale) {
  AreaCell c;
  const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
  const double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
  int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
  sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
  sx1 = sx1 < sx2 ? sx1 : sx2;
  c.head_si = -1; c.head_w = 0.f;
  if (sx1 - fsx1 > 1e-3) { c.head_si = sx1 - 1; c.head_w = (float)((sx1 - fsx1) / cellWidth); }
  c.sx1 = sx1; c.sx2 = sx2; c.full_w = (float)(1.0 / cellWidth);
  c.tail = fsx2 - sx2 > 1e-3;
  c.tail_w = 0.f;
  if (c.tail) {
    double w = fsx2 - sx2;
    w = w > 1. ? 1. : w;
    w = w > cellWidth ? cellWidth : w;
    c.tail_w = (float)(w / cellWidth);
  }
  return c;
}

__global__ __launch_bounds__(256) void k_resize_u8(ResizeArgsK a) {
  const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
  if (dx >= a.dw) return;
  const uint8_t* __restrict__ src = st_gl(a.src[blockIdx.z]);
  uint8_t* __restrict__ D = st_gl(a.dst[blockIdx.z]) + ((size_t)dy * a.dw + dx) * a.cn;
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
  } else if (a.mode == RS_CUBIC) {
    float fx = (float)((dx + 0.5) * a.scale_x - 0.5);
    const int sx = (int)floorf(fx);
    fx -= sx;
    float fy = (float)((dy + 0.5) * a.scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    float cx[4], cy[4];
    rs_cubic(fx, cx);
    rs_cubic(fy, cy);
    int ax[4], by[4], xs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ax[k] = rs_coef(cx[k] * 2048);
      by[k] = rs_coef(cy[k] * 2048);
      const int xx = sx - 1 + k;
      xs[k] = xx < 0 ? 0 : (xx > a.sw - 1 ? a.sw - 1 : xx);  // columns outside the row: edge pixel
    }
    for (int c = 0; c < cn; ++c) {
      int v = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int yy = sy - 1 + k;
        const uint8_t* S = src + (size_t)(yy < 0 ? 0 : (yy > a.sh - 1 ? a.sh - 1 : yy)) * srow;
        const int r = S[(size_t)xs[0] * cn + c] * ax[0] + S[(size_t)xs[1] * cn + c] * ax[1] +
                      S[(size_t)xs[2] * cn + c] * ax[2] + S[(size_t)xs[3] * cn + c] * ax[3];
        v += r * by[k];
      }
      const int o = (v + (1 << 21)) >> 22;
      D[c] = (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
    }
  } else if (a.mode == RS_LANCZOS4) {
    // HResizeLanczos4 + VResizeLanczos4 for 8-bit data: 8 x 8 taps from (sx - 3, sy - 3), the edge
    // pixel replicated outside the image, (v + 2^21) >> 22 saturated
    const int sx = a.xofs[dx], sy = a.yofs[dy];
    int ax[8], by[8], xs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ax[k] = a.ialpha[8 * dx + k];
      by[k] = a.ibeta[8 * dy + k];
      const int xx = sx - 3 + k;
      xs[k] = (xx < 0 ? 0 : (xx > a.sw - 1 ? a.sw - 1 : xx)) * cn;
    }
    for (int c = 0; c < cn; ++c) {
      int v = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int yy = sy - 3 + k;
        const uint8_t* S = src + (size_t)(yy < 0 ? 0 : (yy > a.sh - 1 ? a.sh - 1 : yy)) * srow + c;
        int r = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) r += S[xs[j]] * ax[j];
        v += r * by[k];
      }
      const int o = (v + (1 << 21)) >> 22;
      D[c] = (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
    }
  } else if (a.mode == RS_AREA_INT) {
    const float scale = 1.f / (a.iscale_x * a.iscale_y);
    for (int c = 0; c < cn; ++c) {
      int sum = 0;
      for (int yy = 0; yy < a.iscale_y; ++yy) {
        const uint8_t* S = src + (size_t)(dy * a.iscale_y + yy) * srow + (size_t)(dx * a.iscale_x) * cn + c;
        for (int xx = 0; xx < a.iscale_x; ++xx) sum += S[(size_t)xx * cn];
      }
      D[c] = rs_sat_float(sum * scale);
    }
  } else if (a.mode == RS_AREA) {
    // ResizeArea_Invoker: per source row a float row sum over the cell's columns (head, full cells,
    // tail, in that order, starting from 0), rows accumulated as beta * rowsum in the same order
    const AreaCell cx = rs_area_cell(dx, a.sw, a.scale_x), cy = rs_area_cell(dy, a.sh, a.scale_y);
    for (int c = 0; c < cn; ++c) {
      float sum = 0.f;
      auto rowsum = [&](int sy) {
        const uint8_t* S = src + (size_t)sy * srow + c;
        float b = 0.f;
        if (cx.head_si >= 0) b += S[(size_t)cx.head_si * cn] * cx.head_w;
        for (int sx = cx.sx1; sx < cx.sx2; ++sx) b += S[(size_t)sx * cn] * cx.full_w;
        if (cx.tail) b += S[(size_t)cx.sx2 * cn] * cx.tail_w;
        return b;
      };
      bool first = true;
      auto acc = [&](int sy, float beta) {
        const float t = beta * rowsum(sy);
        sum = first ? t : sum + t;
        first = false;
      };
      if (cy.head_si >= 0) acc(cy.head_si, cy.head_w);
      for (int sy = cy.sx1; sy < cy.sx2; ++sy) acc(sy, cy.full_w);
      if (cy.tail) acc(cy.sx2, cy.tail_w);
      D[c] = rs_sat_float(sum);
    }
  } else {
    float fx, fy;
    int sx, sy;
    if (a.mode == RS_LINEAR_AREA) {  // INTER_AREA when an axis is enlarged: cell-aligned weights
      sx = (int)floor(dx * a.scale_x);
      fx = (float)((dx + 1) - (sx + 1) * a.inv_scale_x);
      fx = fx <= 0 ? 0.f : fx - floorf(fx);
      sy = (int)floor(dy * a.scale_y);
      fy = (float)((dy + 1) - (sy + 1) * a.inv_scale_y);
      fy = fy <= 0 ? 0.f : fy - floorf(fy);
    } else {
      fx = (float)((dx + 0.5) * a.scale_x - 0.5);
      sx = (int)floorf(fx);
      fx -= sx;
      fy = (float)((dy + 0.5) * a.scale_y - 0.5);
      sy = (int)floorf(fy);
      fy -= sy;
    }
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= a.sw - 1) { fx = 0; sx = a.sw - 1; }
    const int a0 = rs_coef((1.f - fx) * 2048), a1 = rs_coef(fx * 2048);
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

// INTER_LINEAR on 3-channel frames, the op's default and commonest case, restructured around what k_resize_u8 spends its
// time on (per-pixel double-precision coordinates and byte stores): a thread owns FOUR consecutive output columns for a
// strip of RL_ROWS rows -- the column coordinates and weights are computed once per strip, the row's once per 4 pixels --
// and stores each row's 12 bytes as three dwords.  Same arithmetic as the generic kernel, value for value.
// (Rows of 3 * dw bytes start at any byte: the stores are unaligned dwords; a row's last, partial group goes out bytewise.)
constexpr int RL_ROWS = 4;
// AREA_UP: INTER_AREA when an axis is enlarged = the same two-tap filter with cell-aligned weights (RS_LINEAR_AREA)
template <bool AREA_UP>
__global__ __launch_bounds__(256) void k_resize_linear_c3_v4(ResizeArgsK a) {
  const int g = blockIdx.x * 256 + threadIdx.x;  // group of 4 output columns
  if (4 * g >= a.dw) return;
  const int npx = min(4, a.dw - 4 * g);          // the last group of a row may be partial
  const uint8_t* __restrict__ src = st_gl(a.src[blockIdx.z]);
  unsigned* __restrict__ dst = reinterpret_cast<unsigned*>(st_gl(a.dst[blockIdx.z]));
  const size_t srow = (size_t)a.sw * 3;
  int sxo[4], a0[4], a1[4];
  bool two[4], wide[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int dx = min(4 * g + p, a.dw - 1);
    float fx;
    int sx;
    if (AREA_UP) {
      sx = (int)floor(dx * a.scale_x);
      fx = (float)((dx + 1) - (sx + 1) * a.inv_scale_x);
      fx = fx <= 0 ? 0.f : fx - floorf(fx);
    } else {
      fx = (float)((dx + 0.5) * a.scale_x - 0.5);
      sx = (int)floorf(fx);
      fx -= sx;
    }
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= a.sw - 1) { fx = 0; sx = a.sw - 1; }
    a0[p] = rs_coef((1.f - fx) * 2048); a1[p] = rs_coef(fx * 2048);
    two[p] = sx + 1 < a.sw;  // the right edge takes a single tap * 2048
    sxo[p] = sx * 3;
    wide[p] = (size_t)sx * 3 + 8 <= srow;
  }
  const int dy0 = blockIdx.y * RL_ROWS;
  for (int dy = dy0; dy < min(a.dh, dy0 + RL_ROWS); ++dy) {
    float fy;
    int sy;
    if (AREA_UP) {
      sy = (int)floor(dy * a.scale_y);
      fy = (float)((dy + 1) - (sy + 1) * a.inv_scale_y);
      fy = fy <= 0 ? 0.f : fy - floorf(fy);
    } else {
      fy = (float)((dy + 0.5) * a.scale_y - 0.5);
      sy = (int)floorf(fy);
      fy -= sy;
    }
    const int b0 = rs_coef((1.f - fy) * 2048), b1 = rs_coef(fy * 2048);
    const int y0 = sy < 0 ? 0 : (sy > a.sh - 1 ? a.sh - 1 : sy);
    const int y1 = sy + 1 < 0 ? 0 : (sy + 1 > a.sh - 1 ? a.sh - 1 : sy + 1);
    const uint8_t* __restrict__ R0 = src + (size_t)y0 * srow;
    const uint8_t* __restrict__ R1 = src + (size_t)y1 * srow;
    unsigned out[3] = {0u, 0u, 0u};
    typedef unsigned u32u __attribute__((aligned(1)));
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      // the two source pixels of a row are 6 contiguous bytes at any byte offset: two unaligned dword loads instead of
      // six byte loads (vector-memory instructions are what this kernel is short of), except where 8 bytes would run
      // past the end of the row
      int t0[6], t1[6];
      if (wide[p]) {
        const unsigned l0 = *reinterpret_cast<const u32u*>(R0 + sxo[p]), h0 = *reinterpret_cast<const u32u*>(R0 + sxo[p] + 4);
        const unsigned l1 = *reinterpret_cast<const u32u*>(R1 + sxo[p]), h1 = *reinterpret_cast<const u32u*>(R1 + sxo[p] + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { t0[k] = (l0 >> (8 * k)) & 0xff; t1[k] = (l1 >> (8 * k)) & 0xff; }
        t0[4] = h0 & 0xff; t0[5] = (h0 >> 8) & 0xff; t1[4] = h1 & 0xff; t1[5] = (h1 >> 8) & 0xff;
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { t0[k] = R0[sxo[p] + k]; t1[k] = R1[sxo[p] + k]; }
#pragma unroll
        for (int k = 3; k < 6; ++k) { t0[k] = two[p] ? R0[sxo[p] + k] : 0; t1[k] = two[p] ? R1[sxo[p] + k] : 0; }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int r0 = two[p] ? t0[c] * a0[p] + t0[3 + c] * a1[p] : t0[c] * 2048;
        const int r1 = two[p] ? t1[c] * a0[p] + t1[3 + c] * a1[p] : t1[c] * 2048;
        const unsigned v = (unsigned)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2) & 0xffu;
        out[(3 * p + c) >> 2] |= v << (8 * ((3 * p + c) & 3));
      }
    }
    uint8_t* ob = reinterpret_cast<uint8_t*>(dst) + ((size_t)dy * a.dw + 4 * g) * 3;  // any byte when 3 * dw % 4 != 0
    if (npx == 4) {
      u32u* o = reinterpret_cast<u32u*>(ob);
      o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
    } else {
      for (int k = 0; k < 3 * npx; ++k) ob[k] = (uint8_t)(out[k >> 2] >> (8 * (k & 3)));
    }
  }
}

// The exact 2 x 2 decimation (INTER_AREA, and INTER_LINEAR's reroute to it) on 3-channel frames: four output pixels per
// thread = 24 contiguous source bytes in each of two rows (six unaligned dword loads per row instead of 24 byte loads),
// twelve output bytes as three unaligned dword stores.  (v00 + v01 + v10 + v11 + 2) >> 2 as in k_resize_u8.
__global__ __launch_bounds__(256) void k_resize_area2_c3_v4(ResizeArgsK a) {
  typedef unsigned u32u __attribute__((aligned(1)));
  const int g = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
  if (4 * g >= a.dw) return;
  const uint8_t* __restrict__ src = st_gl(a.src[blockIdx.z]);
  uint8_t* ob = st_gl(a.dst[blockIdx.z]) + ((size_t)dy * a.dw + 4 * g) * 3;
  const size_t srow = (size_t)a.sw * 3;
  const uint8_t* S0 = src + (size_t)(2 * dy) * srow + (size_t)(8 * g) * 3;
  const uint8_t* S1 = S0 + srow;
  if (4 * g + 4 <= a.dw) {
    unsigned w0[6], w1[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { w0[k] = reinterpret_cast<const u32u*>(S0)[k]; w1[k] = reinterpret_cast<const u32u*>(S1)[k]; }
    unsigned out[3] = {0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int i0 = 6 * p + c, i1 = 6 * p + 3 + c;  // bytes of the two source columns
        const unsigned v = (((w0[i0 >> 2] >> (8 * (i0 & 3))) & 0xffu) + ((w0[i1 >> 2] >> (8 * (i1 & 3))) & 0xffu) +
                            ((w1[i0 >> 2] >> (8 * (i0 & 3))) & 0xffu) + ((w1[i1 >> 2] >> (8 * (i1 & 3))) & 0xffu) + 2u) >> 2;
        out[(3 * p + c) >> 2] |= v << (8 * ((3 * p + c) & 3));
      }
    u32u* o = reinterpret_cast<u32u*>(ob);
    o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
  } else {
    for (int p = 0; 4 * g + p < a.dw; ++p)
      for (int c = 0; c < 3; ++c)
        ob[3 * p + c] = (uint8_t)((S0[6 * p + c] + S0[6 * p + 3 + c] + S1[6 * p + c] + S1[6 * p + 3 + c] + 2) >> 2);
  }
}

// INTER_NEAREST and INTER_CUBIC on 3-channel frames with the same recipe as k_resize_linear_c3_v4: four output columns x
// RL_ROWS rows per thread, column positions / weights once per strip, a tap's pixels through unaligned dword loads, twelve
// output bytes as three unaligned dword stores.  Arithmetic as in k_resize_u8, value for value.
__global__ __launch_bounds__(256) void k_resize_nearest_c3_v4(ResizeArgsK a) {
  typedef unsigned u32u __attribute__((aligned(1)));
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (4 * g >= a.dw) return;
  const int npx = min(4, a.dw - 4 * g);
  const uint8_t* __restrict__ src = st_gl(a.src[blockIdx.z]);
  uint8_t* __restrict__ dstb = st_gl(a.dst[blockIdx.z]);
  const size_t srow = (size_t)a.sw * 3;
  int sxo[4];
  bool wide[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int dx = min(4 * g + p, a.dw - 1);
    int sx = (int)floor(dx * a.scale_x);
    sx = sx < a.sw - 1 ? sx : a.sw - 1;
    sxo[p] = sx * 3;
    wide[p] = (size_t)sx * 3 + 4 <= srow;
  }
  const int dy0 = blockIdx.y * RL_ROWS;
  for (int dy = dy0; dy < min(a.dh, dy0 + RL_ROWS); ++dy) {
    int sy = (int)floor(dy * a.scale_y);
    sy = sy < a.sh - 1 ? sy : a.sh - 1;
    const uint8_t* __restrict__ R = src + (size_t)sy * srow;
    unsigned out[3] = {0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      unsigned v;
      if (wide[p]) v = *reinterpret_cast<const u32u*>(R + sxo[p]) & 0xffffffu;
      else v = (unsigned)R[sxo[p]] | ((unsigned)R[sxo[p] + 1] << 8) | ((unsigned)R[sxo[p] + 2] << 16);
      // the pixel's 3 bytes land at output byte 3 p
      if (p == 0) out[0] |= v;
      else if (p == 1) { out[0] |= v << 24; out[1] |= v >> 8; }
      else if (p == 2) { out[1] |= v << 16; out[2] |= v >> 16; }
      else out[2] |= v << 8;
    }
    uint8_t* ob = dstb + ((size_t)dy * a.dw + 4 * g) * 3;
    if (npx == 4) {
      u32u* o = reinterpret_cast<u32u*>(ob);
      o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
    } else {
      for (int k = 0; k < 3 * npx; ++k) ob[k] = (uint8_t)(out[k >> 2] >> (8 * (k & 3)));
    }
  }
}

__global__ __launch_bounds__(256) void k_resize_cubic_c3_v4(ResizeArgsK a) {
  typedef unsigned u32u __attribute__((aligned(1)));
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (4 * g >= a.dw) return;
  const int npx = min(4, a.dw - 4 * g);
  const uint8_t* __restrict__ src = st_gl(a.src[blockIdx.z]);
  uint8_t* __restrict__ dstb = st_gl(a.dst[blockIdx.z]);
  const size_t srow = (size_t)a.sw * 3;
  int ax[4][4], xs[4][4];
  bool run[4];  // the four tap columns are consecutive and 12 bytes from the first stay inside the row
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int dx = min(4 * g + p, a.dw - 1);
    float fx = (float)((dx + 0.5) * a.scale_x - 0.5);
    const int sx = (int)floorf(fx);
    fx -= sx;
    float cx[4];
    rs_cubic(fx, cx);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ax[p][k] = rs_coef(cx[k] * 2048);
      const int xx = sx - 1 + k;
      xs[p][k] = (xx < 0 ? 0 : (xx > a.sw - 1 ? a.sw - 1 : xx)) * 3;  // columns outside the row: edge pixel
    }
    run[p] = sx - 1 >= 0 && sx + 2 <= a.sw - 1;
  }
  const int dy0 = blockIdx.y * RL_ROWS;
  for (int dy = dy0; dy < min(a.dh, dy0 + RL_ROWS); ++dy) {
    float fy = (float)((dy + 0.5) * a.scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    float cy[4];
    rs_cubic(fy, cy);
    int by[4];
    const uint8_t* R[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      by[k] = rs_coef(cy[k] * 2048);
      const int yy = sy - 1 + k;
      R[k] = src + (size_t)(yy < 0 ? 0 : (yy > a.sh - 1 ? a.sh - 1 : yy)) * srow;
    }
    unsigned out[3] = {0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int v[3] = {0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int t[12];  // the row's four tap pixels, 3 bytes each
        if (run[p]) {
          const unsigned w0 = *reinterpret_cast<const u32u*>(R[k] + xs[p][0]), w1 = *reinterpret_cast<const u32u*>(R[k] + xs[p][0] + 4),
                         w2 = *reinterpret_cast<const u32u*>(R[k] + xs[p][0] + 8);
#pragma unroll
          for (int j = 0; j < 4; ++j) { t[j] = (w0 >> (8 * j)) & 0xff; t[4 + j] = (w1 >> (8 * j)) & 0xff; t[8 + j] = (w2 >> (8 * j)) & 0xff; }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) t[3 * q + c] = R[k][xs[p][q] + c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int r = t[c] * ax[p][0] + t[3 + c] * ax[p][1] + t[6 + c] * ax[p][2] + t[9 + c] * ax[p][3];
          v[c] += r * by[k];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        int o = (v[c] + (1 << 21)) >> 22;
        // Keeps hipcc (ROCm 7.2) from fusing shift + clamp + pack of two values into v_ashr_pk_u8_i32: on the GPU that
        // instruction's result carries bits above 15, which the v_lshl_or that assembles the dword then ORs into bytes 2
        // and 3 (measured: bytes 0 and 1 of every output dword right, 2 and 3 wrong).
        asm volatile("" : "+v"(o));
        out[(3 * p + c) >> 2] |= ((unsigned)(o < 0 ? 0 : (o > 255 ? 255 : o)) & 0xffu) << (8 * ((3 * p + c) & 3));
      }
    }
    uint8_t* ob = dstb + ((size_t)dy * a.dw + 4 * g) * 3;
    if (npx == 4) {
      u32u* o = reinterpret_cast<u32u*>(ob);
      o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
    } else {
      for (int k = 0; k < 3 * npx; ++k) ob[k] = (uint8_t)(out[k >> 2] >> (8 * (k & 3)));
    }
  }
}

// INTER_AREA with fractional cells (shrinking by a non-integer factor) on 3-channel frames: four output columns x RL_ROWS rows
// per thread.  A column's cell (head, full cells, tail: at most AR_SPAN source columns here, else the generic kernel) is
// worked out once per strip as a first column and AR_SPAN (4 or 8) weights in ascending column order -- absent ones are 0, and
// adding 0.f is exact, so the row sum b = ((t0 + t1) + t2) + ... is ResizeArea_Invoker's, bit for bit; the rows are
// accumulated as in k_resize_u8 (beta * rowsum, head / full / tail order).  Source pixels through unaligned dword loads.
template <int AR_SPAN>
__global__ __launch_bounds__(256) void k_resize_area_c3_v4(ResizeArgsK a) {
  typedef unsigned u32u __attribute__((aligned(1)));
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (4 * g >= a.dw) return;
  const int npx = min(4, a.dw - 4 * g);
  const uint8_t* __restrict__ src = st_gl(a.src[blockIdx.z]);
  uint8_t* __restrict__ dstb = st_gl(a.dst[blockIdx.z]);
  const size_t srow = (size_t)a.sw * 3;
  int first[4];
  float wx[4][AR_SPAN];
  bool wide[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const AreaCell c = rs_area_cell(min(4 * g + p, a.dw - 1), a.sw, a.scale_x);
    first[p] = c.head_si >= 0 ? c.head_si : c.sx1;
#pragma unroll
    for (int j = 0; j < AR_SPAN; ++j) {
      const int sx = first[p] + j;
      wx[p][j] = (c.head_si >= 0 && sx == c.head_si) ? c.head_w : ((sx >= c.sx1 && sx < c.sx2) ? c.full_w : ((c.tail && sx == c.sx2) ? c.tail_w : 0.f));
    }
    wide[p] = (size_t)first[p] * 3 + 3 * AR_SPAN <= srow;
  }
  const int dy0 = blockIdx.y * RL_ROWS;
  for (int dy = dy0; dy < min(a.dh, dy0 + RL_ROWS); ++dy) {
    const AreaCell cy = rs_area_cell(dy, a.sh, a.scale_y);
    float sum[4][3];
    bool started = false;
    auto acc = [&](int sy, float beta) {
      const uint8_t* __restrict__ R = src + (size_t)sy * srow;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        int t[3 * AR_SPAN];
        if (wide[p]) {
#pragma unroll
          for (int q = 0; q < 3 * AR_SPAN / 4; ++q) {
            const unsigned wv = *reinterpret_cast<const u32u*>(R + (size_t)first[p] * 3 + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) t[4 * q + j] = (wv >> (8 * j)) & 0xff;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 3 * AR_SPAN; ++j) {
            const size_t o = (size_t)first[p] * 3 + j;
            t[j] = o < srow ? R[o] : 0;  // columns past the row carry weight 0
          }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float b = (float)t[c] * wx[p][0];
#pragma unroll
          for (int j = 1; j < AR_SPAN; ++j) b += (float)t[3 * j + c] * wx[p][j];
          const float v = beta * b;
          sum[p][c] = started ? sum[p][c] + v : v;
        }
      }
      started = true;
    };
    if (cy.head_si >= 0) acc(cy.head_si, cy.head_w);
    for (int sy = cy.sx1; sy < cy.sx2; ++sy) acc(sy, cy.full_w);
    if (cy.tail) acc(cy.sx2, cy.tail_w);
    unsigned out[3] = {0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < 3; ++c) out[(3 * p + c) >> 2] |= ((unsigned)rs_sat_float(started ? sum[p][c] : 0.f) & 0xffu) << (8 * ((3 * p + c) & 3));
    uint8_t* ob = dstb + ((size_t)dy * a.dw + 4 * g) * 3;
    if (npx == 4) {
      u32u* o = reinterpret_cast<u32u*>(ob);
      o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
    } else {
      for (int k = 0; k < 3 * npx; ++k) ob[k] = (uint8_t)(out[k >> 2] >> (8 * (k & 3)));
    }
  }
}

// INTER_AREA with integer cells (shrinking by whole factors, here up to 4 columns per cell) on 3-channel frames: the
// 4 * iscale_x source pixels of four output pixels are one contiguous run per source row, read as unaligned dwords;
// integer sums, then saturate_cast<uchar>(sum * (1 / area)) as in k_resize_u8.
template <int ISX>
__global__ __launch_bounds__(256) void k_resize_area_int_c3_v4(ResizeArgsK a) {
  typedef unsigned u32u __attribute__((aligned(1)));
  const int g = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
  if (4 * g >= a.dw) return;
  const int npx = min(4, a.dw - 4 * g);
  const uint8_t* __restrict__ src = st_gl(a.src[blockIdx.z]);
  uint8_t* ob = st_gl(a.dst[blockIdx.z]) + ((size_t)dy * a.dw + 4 * g) * 3;
  const size_t srow = (size_t)a.sw * 3;
  constexpr int isx = ISX;                          // 2 .. 4 columns per cell
  const size_t base = (size_t)(4 * g) * isx * 3;    // first source byte of the run in a row
  constexpr int nbytes = 12 * isx;                  // 24, 36 or 48
  const bool full = npx == 4 && base + (size_t)nbytes <= srow;
  int sum[4][3];
#pragma unroll
  for (int p = 0; p < 4; ++p) sum[p][0] = sum[p][1] = sum[p][2] = 0;
  for (int yy = 0; yy < a.iscale_y; ++yy) {
    const uint8_t* __restrict__ R = src + (size_t)(dy * a.iscale_y + yy) * srow + base;
    if (full) {
      unsigned wv[3 * isx];
#pragma unroll
      for (int q = 0; q < 3 * isx; ++q) wv[q] = reinterpret_cast<const u32u*>(R)[q];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int xx = 0; xx < isx; ++xx)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int j = (p * isx + xx) * 3 + c;  // byte of the run (compile-time)
            sum[p][c] += (int)((wv[j >> 2] >> (8 * (j & 3))) & 0xffu);
          }
    } else {
      for (int p = 0; p < npx; ++p)
        for (int xx = 0; xx < isx; ++xx)
          for (int c = 0; c < 3; ++c) sum[p][c] += R[(p * isx + xx) * 3 + c];
    }
  }
  const float scale = 1.f / (a.iscale_x * a.iscale_y);
  unsigned out[3] = {0u, 0u, 0u};
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(3 * p + c) >> 2] |= ((unsigned)rs_sat_float(sum[p][c] * scale) & 0xffu) << (8 * ((3 * p + c) & 3));
  if (npx == 4) {
    u32u* o = reinterpret_cast<u32u*>(ob);
    o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
  } else {
    for (int k = 0; k < 3 * npx; ++k) ob[k] = (uint8_t)(out[k >> 2] >> (8 * (k & 3)));
  }
}

// INTER_LANCZOS4 on 3-channel frames: one output pixel per thread as in k_resize_u8, but a source row's eight tap pixels
// (24 contiguous bytes when none of them is clamped at an edge) come in as six unaligned dword loads instead of 24 byte loads.
__global__ __launch_bounds__(256) void k_resize_lanczos4_c3(ResizeArgsK a) {
  typedef unsigned u32u __attribute__((aligned(1)));
  const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
  if (dx >= a.dw) return;
  const uint8_t* __restrict__ src = st_gl(a.src[blockIdx.z]);
  uint8_t* __restrict__ D = st_gl(a.dst[blockIdx.z]) + ((size_t)dy * a.dw + dx) * 3;
  const size_t srow = (size_t)a.sw * 3;
  const int sx = a.xofs[dx], sy = a.yofs[dy];
  int ax[8], xs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    ax[k] = a.ialpha[8 * dx + k];
    const int xx = sx - 3 + k;
    xs[k] = (xx < 0 ? 0 : (xx > a.sw - 1 ? a.sw - 1 : xx)) * 3;
  }
  const bool run = sx - 3 >= 0 && sx + 4 <= a.sw - 1;
  int v[3] = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int yy = sy - 3 + k;
    const uint8_t* __restrict__ S = src + (size_t)(yy < 0 ? 0 : (yy > a.sh - 1 ? a.sh - 1 : yy)) * srow;
    const int by = a.ibeta[8 * dy + k];
    int r[3] = {0, 0, 0};
    if (run) {
      unsigned w[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) w[q] = *reinterpret_cast<const u32u*>(S + xs[0] + 4 * q);
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] += (int)((w[(3 * j + c) >> 2] >> (8 * ((3 * j + c) & 3))) & 0xffu) * ax[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] += S[xs[j] + c] * ax[j];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] += r[c] * by;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int o = (v[c] + (1 << 21)) >> 22;
    D[c] = (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o));
  }
}

// ---- ConvertColor ---------------------------------------------------------------------------------
struct CvtArgsK {
  const uint8_t* const* src;
  uint8_t* const* dst;
  long long npix;
  int code;                // cv::ColorConversionCodes value
  int cb, cg, cr, rnd, shift, bi;  // gray weights (bi = byte holding blue)
  // layout family (alpha channel, 16-bit packed pixels): source / destination kind and the byte that holds blue
  int layout, sk, sbi, dk, dbi;
  int scn, dcn;            // channels per source / destination pixel
};

// pixel kinds of the layout family
enum { PK_C3 = 0, PK_C4 = 1, PK_GRAY = 2, PK_565 = 3, PK_555 = 4 };

// cv::cvtColor codes 0..3, 5, 9..31: the conversions that only move, drop, add or pack channels
// (RGB2RGB<uchar>, Gray2RGB, RGB2Gray on 4 channels, RGB2RGB5x5, RGB5x52RGB, Gray2RGB5x5, RGB5x52Gray)
struct CvtLayout { int sk, sbi, dk, dbi; };
__host__ __device__ inline bool cvt_layout_of(int code, CvtLayout* l) {
  switch (code) {
    case 0: *l = {PK_C3, 0, PK_C4, 0}; return true;   // BGR2BGRA = RGB2RGBA
    case 1: *l = {PK_C4, 0, PK_C3, 0}; return true;   // BGRA2BGR = RGBA2RGB
    case 2: *l = {PK_C3, 0, PK_C4, 2}; return true;   // BGR2RGBA = RGB2BGRA
    case 3: *l = {PK_C4, 0, PK_C3, 2}; return true;   // RGBA2BGR = BGRA2RGB
    case 5: *l = {PK_C4, 0, PK_C4, 2}; return true;   // BGRA2RGBA = RGBA2BGRA
    case 9: *l = {PK_GRAY, 0, PK_C4, 0}; return true;  // GRAY2BGRA = GRAY2RGBA
    case 10: *l = {PK_C4, 0, PK_GRAY, 0}; return true;  // BGRA2GRAY
    case 11: *l = {PK_C4, 2, PK_GRAY, 0}; return true;  // RGBA2GRAY
    default: break;
  }
  if (code >= 12 && code <= 31) {
    const int k = code >= 22 ? PK_555 : PK_565, c = code >= 22 ? code - 22 : code - 12;
    switch (c) {
      case 0: *l = {PK_C3, 0, k, 0}; return true;      // BGR2BGR5x5
      case 1: *l = {PK_C3, 2, k, 0}; return true;      // RGB2BGR5x5
      case 2: *l = {k, 0, PK_C3, 0}; return true;      // BGR5x52BGR
      case 3: *l = {k, 0, PK_C3, 2}; return true;      // BGR5x52RGB
      case 4: *l = {PK_C4, 0, k, 0}; return true;      // BGRA2BGR5x5
      case 5: *l = {PK_C4, 2, k, 0}; return true;      // RGBA2BGR5x5
      case 6: *l = {k, 0, PK_C4, 0}; return true;      // BGR5x52BGRA
      case 7: *l = {k, 0, PK_C4, 2}; return true;      // BGR5x52RGBA
      case 8: *l = {PK_GRAY, 0, k, 0}; return true;    // GRAY2BGR5x5
      case 9: *l = {k, 0, PK_GRAY, 0}; return true;    // BGR5x52GRAY
    }
  }
  return false;
}
__host__ __device__ inline int cvt_kind_channels(int k) { return k == PK_C3 ? 3 : (k == PK_C4 ? 4 : (k == PK_GRAY ? 1 : 2)); }

__device__ __forceinline__ bool cvt_is_to_hsv(int code) {
  return code == ST_COLOR_BGR2HSV || code == ST_COLOR_RGB2HSV || code == ST_COLOR_BGR2HSV_FULL || code == ST_COLOR_RGB2HSV_FULL;
}

__device__ __forceinline__ int cvt_sat(int v) { return min(max(v, 0), 255); }

// One pixel, channel bytes in registers: s[0 .. scn-1] -> d[0 .. dcn-1].  (No indexing by run-time values: the two
// front-ends below unroll over pixels and channels, so s and d stay in VGPRs.)
__device__ __forceinline__ void cvt_pixel(const CvtArgsK& a, const int* sdiv, const int* hdiv, const int s[4], int d[4]) {
  if (a.layout) {
    // decode to (b, g, r, alpha), encode in the destination kind
    int b, g, r, al = 255;
    bool packed_src = false;
    if (a.sk == PK_C3 || a.sk == PK_C4) {
      b = a.sbi ? s[2] : s[0]; g = s[1]; r = a.sbi ? s[0] : s[2];
      if (a.sk == PK_C4) al = s[3];
    } else if (a.sk == PK_GRAY) {
      b = g = r = s[0];
    } else {
      const unsigned tt = (unsigned)s[0] | ((unsigned)s[1] << 8);
      packed_src = true;
      if (a.sk == PK_565) { b = (tt << 3) & 0xff; g = (tt >> 3) & 0xfc; r = (tt >> 8) & 0xf8; }
      else { b = (tt << 3) & 0xf8; g = (tt >> 2) & 0xf8; r = (tt >> 7) & 0xf8; al = (tt & 0x8000) ? 255 : 0; }
    }
    if (a.dk == PK_C3 || a.dk == PK_C4) {
      d[0] = a.dbi ? r : b; d[1] = g; d[2] = a.dbi ? b : r; d[3] = al;
    } else if (a.dk == PK_GRAY) {
      // from 4 channels: RGB2Gray<uchar> (the op's gray table); from packed pixels: RGB5x52Gray, 14-bit weights
      d[0] = packed_src ? (b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14 : (b * a.cb + g * a.cg + r * a.cr + a.rnd) >> a.shift;
    } else {
      unsigned tt;
      if (a.dk == PK_565) tt = (unsigned)(b >> 3) | ((unsigned)(g & ~3) << 3) | ((unsigned)(r & ~7) << 8);
      else tt = (unsigned)(b >> 3) | ((unsigned)(g & ~7) << 2) | ((unsigned)(r & ~7) << 7) | ((a.sk == PK_C4 && al) ? 0x8000u : 0u);
      d[0] = (int)(tt & 0xff); d[1] = (int)(tt >> 8);
    }
  } else if (a.code >= 32 && a.code <= 35) {
    // RGB2XYZ_i<uchar> / XYZ2RGB_i<uchar>: sRGB <-> XYZ (D65) matrices in 12-bit fixed point, CV_DESCALE, saturate
    const int p0 = s[0], p1 = s[1], p2 = s[2];
    if (a.code <= 33) {
      const int b = a.code == 32 ? p0 : p2, g = p1, r = a.code == 32 ? p2 : p0;
      d[0] = cvt_sat((r * 1689 + g * 1465 + b * 739 + (1 << 11)) >> 12);
      d[1] = cvt_sat((r * 871 + g * 2929 + b * 296 + (1 << 11)) >> 12);
      d[2] = cvt_sat((r * 79 + g * 488 + b * 3892 + (1 << 11)) >> 12);
    } else {
      const int r = cvt_sat((p0 * 13273 + p1 * -6296 + p2 * -2042 + (1 << 11)) >> 12);
      const int g = cvt_sat((p0 * -3970 + p1 * 7684 + p2 * 170 + (1 << 11)) >> 12);
      const int b = cvt_sat((p0 * 228 + p1 * -836 + p2 * 4331 + (1 << 11)) >> 12);
      d[0] = a.code == 34 ? b : r; d[1] = g; d[2] = a.code == 34 ? r : b;
    }
  } else if (a.code == ST_COLOR_BGR2RGB) {
    d[0] = s[2]; d[1] = s[1]; d[2] = s[0];
  } else if (a.code == ST_COLOR_BGR2GRAY || a.code == ST_COLOR_RGB2GRAY) {
    const int b = a.bi ? s[2] : s[0], g = s[1], r = a.bi ? s[0] : s[2];
    d[0] = (b * a.cb + g * a.cg + r * a.cr + a.rnd) >> a.shift;
  } else if (a.code == ST_COLOR_GRAY2BGR) {
    d[0] = d[1] = d[2] = s[0];
  } else if (a.code == ST_COLOR_BGR2YCrCb || a.code == ST_COLOR_RGB2YCrCb || a.code == ST_COLOR_BGR2YUV || a.code == ST_COLOR_RGB2YUV) {
    // RGB2YCrCb_i<uchar>: 14-bit {R2Y, G2Y, B2Y} = {4899, 9617, 1868}; chroma gains {YCR, YCB} = {11682, 9241} for YCrCb (stored
    // Y, Cr, Cb), {R2V, B2U} = {14369, 8061} for YUV (stored Y, U, V)
    const bool yuv = a.code == ST_COLOR_BGR2YUV || a.code == ST_COLOR_RGB2YUV;
    const bool bgr = a.code == ST_COLOR_BGR2YCrCb || a.code == ST_COLOR_BGR2YUV;
    const int p0 = s[0], p1 = s[1], p2 = s[2];
    const int C0 = bgr ? 1868 : 4899, C2 = bgr ? 4899 : 1868;
    const int Y = (p0 * C0 + p1 * 9617 + p2 * C2 + (1 << 13)) >> 14;
    const int rr = bgr ? p2 : p0, bb = bgr ? p0 : p2;
    const int Cr = ((rr - Y) * (yuv ? 14369 : 11682) + (128 << 14) + (1 << 13)) >> 14;
    const int Cb = ((bb - Y) * (yuv ? 8061 : 9241) + (128 << 14) + (1 << 13)) >> 14;
    d[0] = cvt_sat(Y); d[1] = cvt_sat(yuv ? Cb : Cr); d[2] = cvt_sat(yuv ? Cr : Cb);
  } else if (a.code == ST_COLOR_YCrCb2BGR || a.code == ST_COLOR_YCrCb2RGB || a.code == ST_COLOR_YUV2BGR || a.code == ST_COLOR_YUV2RGB) {
    // YCrCb2RGB_i<uchar>: {CR2R, CR2G, CB2G, CB2B} = {22987, -11698, -5636, 29049}; YUV: {V2R, V2G, U2G, U2B} = {18678, -9519, -6472, 33292}
    const bool yuv = a.code == ST_COLOR_YUV2BGR || a.code == ST_COLOR_YUV2RGB;
    const bool bgr = a.code == ST_COLOR_YCrCb2BGR || a.code == ST_COLOR_YUV2BGR;
    const int Y = s[0], Cr = (yuv ? s[2] : s[1]) - 128, Cb = (yuv ? s[1] : s[2]) - 128;
    const int b = cvt_sat(Y + ((Cb * (yuv ? 33292 : 29049) + (1 << 13)) >> 14));
    const int g = cvt_sat(Y + ((Cb * (yuv ? -6472 : -5636) + Cr * (yuv ? -9519 : -11698) + (1 << 13)) >> 14));
    const int r = cvt_sat(Y + ((Cr * (yuv ? 18678 : 22987) + (1 << 13)) >> 14));
    d[0] = bgr ? b : r; d[1] = g; d[2] = bgr ? r : b;
  } else if (a.code == ST_COLOR_HSV2BGR || a.code == ST_COLOR_HSV2RGB || a.code == ST_COLOR_HSV2BGR_FULL || a.code == ST_COLOR_HSV2RGB_FULL) {
    // HSV2RGB_b: bytes -> (h, s/255, v/255) -> HSV2RGB_native in float -> saturate_cast<uchar>(x * 255)
    const bool bgr = a.code == ST_COLOR_HSV2BGR || a.code == ST_COLOR_HSV2BGR_FULL;
    const float hscale = (a.code == ST_COLOR_HSV2BGR || a.code == ST_COLOR_HSV2RGB) ? 6.f / 180 : 6.f / 255;
    float hh = (float)s[0];
    const float ss = (float)s[1] * (1.f / 255.f), vv = (float)s[2] * (1.f / 255.f);
    float b, g, r;
    if (ss == 0) {
      b = g = r = vv;
    } else {
      hh *= hscale;
      hh = fmodf(hh, 6.f);
      int sector = (int)floorf(hh);
      hh -= sector;
      if ((unsigned)sector >= 6u) { sector = 0; hh = 0.f; }
      const float t0 = vv, t1 = vv * (1.f - ss), t2 = vv * (1.f - ss * hh), t3 = vv * (1.f - ss * (1.f - hh));
      // sector table {{1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}} -> (b, g, r)
      b = sector == 0 || sector == 1 ? t1 : (sector == 2 ? t3 : (sector == 5 ? t2 : t0));
      g = sector == 0 ? t3 : (sector == 1 || sector == 2 ? t0 : (sector == 3 ? t2 : t1));
      r = sector == 0 || sector == 5 ? t0 : (sector == 1 ? t2 : (sector == 4 ? t3 : t1));
    }
    const int bb = rs_sat_float(b * 255.f), gg = rs_sat_float(g * 255.f), rr = rs_sat_float(r * 255.f);
    d[0] = bgr ? bb : rr; d[1] = gg; d[2] = bgr ? rr : bb;
  } else if (a.code == ST_COLOR_BGR2HLS || a.code == ST_COLOR_RGB2HLS || a.code == ST_COLOR_BGR2HLS_FULL || a.code == ST_COLOR_RGB2HLS_FULL) {
    // RGB2HLS_b: bytes / 255 -> RGB2HLS_f in float (scalar formulation) -> H = saturate_cast<uchar>(h * hrange / 360), L, S x 255
    const bool bgr = a.code == ST_COLOR_BGR2HLS || a.code == ST_COLOR_BGR2HLS_FULL;
    const float hscale = ((a.code == ST_COLOR_BGR2HLS || a.code == ST_COLOR_RGB2HLS) ? 180.f : 256.f) / 360.f;
    const float b = (float)(bgr ? s[0] : s[2]) * (1.f / 255.f), g = (float)s[1] * (1.f / 255.f), r = (float)(bgr ? s[2] : s[0]) * (1.f / 255.f);
    float h = 0.f, sat = 0.f;
    const float vmax = fmaxf(r, fmaxf(g, b)), vmin = fminf(r, fminf(g, b));
    float diff = vmax - vmin;
    const float l = (vmax + vmin) * 0.5f;
    if (diff > 1.1920929e-07f) {   // FLT_EPSILON
      sat = l < 0.5f ? diff / (vmax + vmin) : diff / (2.f - vmax - vmin);
      diff = 60.f / diff;
      if (vmax == r) h = (g - b) * diff;
      else if (vmax == g) h = (b - r) * diff + 120.f;
      else h = (r - g) * diff + 240.f;
      if (h < 0.f) h += 360.f;
    }
    d[0] = rs_sat_float(h * hscale); d[1] = rs_sat_float(l * 255.f); d[2] = rs_sat_float(sat * 255.f);
  } else if (a.code == ST_COLOR_HLS2BGR || a.code == ST_COLOR_HLS2RGB || a.code == ST_COLOR_HLS2BGR_FULL || a.code == ST_COLOR_HLS2RGB_FULL) {
    // HLS2RGB_b: bytes -> (h, l/255, s/255) -> HLS2RGB_native in float -> saturate_cast<uchar>(x * 255)
    const bool bgr = a.code == ST_COLOR_HLS2BGR || a.code == ST_COLOR_HLS2BGR_FULL;
    const float hscale = (a.code == ST_COLOR_HLS2BGR || a.code == ST_COLOR_HLS2RGB) ? 6.f / 180 : 6.f / 255;
    float hh = (float)s[0];
    const float l = (float)s[1] * (1.f / 255.f), ss = (float)s[2] * (1.f / 255.f);
    float b, g, r;
    if (ss == 0) {
      b = g = r = l;
    } else {
      const float p2 = l <= 0.5f ? l * (1.f + ss) : l + ss - l * ss;
      const float p1 = 2.f * l - p2;
      hh *= hscale;
      while (hh >= 6.f) hh -= 6.f;   // a byte times 6 / hrange is never negative
      int sector = (int)floorf(hh);
      hh -= sector;
      if ((unsigned)sector >= 6u) { sector = 0; hh = 0.f; }
      const float t0 = p2, t1 = p1, t2 = p1 + (p2 - p1) * (1.f - hh), t3 = p1 + (p2 - p1) * hh;
      // sector table {{1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}} -> (b, g, r)
      b = sector == 0 || sector == 1 ? t1 : (sector == 2 ? t3 : (sector == 5 ? t2 : t0));
      g = sector == 0 ? t3 : (sector == 1 || sector == 2 ? t0 : (sector == 3 ? t2 : t1));
      r = sector == 0 || sector == 5 ? t0 : (sector == 1 ? t2 : (sector == 4 ? t3 : t1));
    }
    const int bb = rs_sat_float(b * 255.f), gg = rs_sat_float(g * 255.f), rr = rs_sat_float(r * 255.f);
    d[0] = bgr ? bb : rr; d[1] = gg; d[2] = bgr ? rr : bb;
  } else {  // BGR2HSV / RGB2HSV, hue range 180 (256 for _FULL)
    const bool bgr = a.code == ST_COLOR_BGR2HSV || a.code == ST_COLOR_BGR2HSV_FULL;
    const int hr = (a.code == ST_COLOR_BGR2HSV || a.code == ST_COLOR_RGB2HSV) ? 180 : 256;
    const int b = bgr ? s[0] : s[2], g = s[1], r = bgr ? s[2] : s[0];
    const int v = max(b, max(g, r)), vmin = min(b, min(g, r));
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int sv = (diff * sdiv[v] + (1 << 11)) >> 12;
    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hh = (hh * hdiv[diff] + (1 << 11)) >> 12;
    hh += hh < 0 ? hr : 0;
    d[0] = cvt_sat(hh); d[1] = sv; d[2] = v;
  }
}

__device__ __forceinline__ void cvt_tables(const CvtArgsK& a, int* sdiv, int* hdiv) {
  if (cvt_is_to_hsv(a.code)) {
    // RGB2HSV_b tables: saturate_cast<int>((255 << 12)/(1.*i)), saturate_cast<int>((hrange << 12)/(6.*i))
    const int t = threadIdx.x;
    const int hr = (a.code == ST_COLOR_BGR2HSV || a.code == ST_COLOR_RGB2HSV) ? 180 : 256;
    sdiv[t] = t ? (int)rint((255 << 12) / (1. * t)) : 0;
    hdiv[t] = t ? (int)rint((hr << 12) / (6. * t)) : 0;
    __syncthreads();
  }
}

// byte-wise front-end: any size, any alignment
__global__ __launch_bounds__(256) void k_cvt_color_u8(CvtArgsK a) {
  __shared__ int sdiv[256], hdiv[256];
  cvt_tables(a, sdiv, hdiv);
  const uint8_t* __restrict__ src = st_gl(st_gl(a.src[blockIdx.y]));
  uint8_t* __restrict__ dst = st_gl(st_gl(a.dst[blockIdx.y]));
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.npix; i += (long long)gridDim.x * 256) {
    int s[4] = {0, 0, 0, 0}, d[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < a.scn) s[k] = src[(size_t)a.scn * i + k];
    cvt_pixel(a, sdiv, hdiv, s, d);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < a.dcn) dst[(size_t)a.dcn * i + k] = (uint8_t)d[k];
  }
}

// PX = 4 or 16 pixels per thread through whole dwords (npix a multiple of PX; frames 4- / 16-byte aligned): PX = 16 moves
// SCN 16-byte loads and DCN 16-byte stores per thread, the lanes of a wave covering one contiguous run of the frame
template <int SCN, int DCN, int PX>
__global__ __launch_bounds__(256) void k_cvt_color_u8_vec(CvtArgsK a) {
  __shared__ int sdiv[256], hdiv[256];
  cvt_tables(a, sdiv, hdiv);
  constexpr int NI = SCN * PX / 4, NO = DCN * PX / 4;  // dwords in / out per thread
  const unsigned* __restrict__ src = reinterpret_cast<const unsigned*>(st_gl(a.src[blockIdx.y]));
  unsigned* __restrict__ dst = reinterpret_cast<unsigned*>(st_gl(a.dst[blockIdx.y]));
  const long long groups = a.npix / PX;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < groups; i += (long long)gridDim.x * 256) {
    unsigned in[NI], out[NO];
    if (PX == 16) {
#pragma unroll
      for (int k = 0; k < NI / 4; ++k) {
        const uint4 v = reinterpret_cast<const uint4*>(src + i * NI)[k];
        in[4 * k] = v.x; in[4 * k + 1] = v.y; in[4 * k + 2] = v.z; in[4 * k + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < NI; ++k) in[k] = src[i * NI + k];
    }
#pragma unroll
    for (int k = 0; k < NO; ++k) out[k] = 0;
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      int s[4] = {0, 0, 0, 0}, d[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < SCN; ++k) s[k] = (int)((in[(p * SCN + k) >> 2] >> (8 * ((p * SCN + k) & 3))) & 0xffu);
      cvt_pixel(a, sdiv, hdiv, s, d);
#pragma unroll
      for (int k = 0; k < DCN; ++k) out[(p * DCN + k) >> 2] |= ((unsigned)d[k] & 0xffu) << (8 * ((p * DCN + k) & 3));
    }
    if (PX == 16) {
#pragma unroll
      for (int k = 0; k < NO / 4; ++k)
        reinterpret_cast<uint4*>(dst + i * NO)[k] = make_uint4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
    } else {
#pragma unroll
      for (int k = 0; k < NO; ++k) dst[i * NO + k] = out[k];
    }
  }
}

// ---- YUV 4:2:0 / 4:2:2 sources (what a video decoder hands out) -> RGB / BGR / RGBA / BGRA / gray ------------------------
// cv::cvtColor codes 90..106 (NV12, NV21, YV12, IYUV = I420; a (3H/2, W) single-channel frame: Y plane, then the chroma
// interleaved (NV12: U V, NV21: V U) or as two quarter planes (I420: U V, YV12: V U)) and 107..124 (UYVY, YUY2, YVYU: a
// (H, W, 2) frame, two pixels per 4 bytes).  OpenCV's YUV420sp2RGB888Invoker / YUV420p2RGB888Invoker / YUV422toRGB888Invoker:
// ITU-R BT.601 in 20-bit fixed point, chroma replicated over its 2x2 (2x1) pixels, no interpolation:
//   y = max(0, Y - 16) * CY;  R = sat((y + (1<<19) + CVR v) >> 20), G = sat((y + (1<<19) + CVG v + CUG u) >> 20),
//   B = sat((y + (1<<19) + CUB u) >> 20),  u = U - 128, v = V - 128,
//   CY, CUB, CUG, CVG, CVR = 1220542, 2116026, -409993, -852492, 1673527 (1.164, 2.018, -0.391, -0.813, 1.596 x 2^20).
struct YuvDesc {
  int kind;   // 0: semi-planar 4:2:0, 1: planar 4:2:0, 2: packed 4:2:2, 3: gray from 4:2:0, 4: gray from 4:2:2
  int bidx;   // byte of the output pixel that holds blue
  int uidx;   // 4:2:0: 0 = U first; 4:2:2: byte of U inside the 4-byte group
  int yidx;   // 4:2:2: byte of the first luma sample inside the group (0 or 1)
  int dcn;    // output channels
};
__host__ __device__ inline bool cvt_yuv_of(int code, YuvDesc* d) {
  if (code >= 90 && code <= 97) {   // RGB_NV12, BGR_NV12, RGB_NV21, BGR_NV21, RGBA_NV12, BGRA_NV12, RGBA_NV21, BGRA_NV21
    const int c = code - 90;
    *d = {0, (c & 1) ? 0 : 2, ((c >> 1) & 1), 0, c >= 4 ? 4 : 3};
    return true;
  }
  if (code >= 98 && code <= 105) {  // RGB_YV12, BGR_YV12, RGB_IYUV, BGR_IYUV, RGBA_YV12, BGRA_YV12, RGBA_IYUV, BGRA_IYUV
    const int c = code - 98;
    *d = {1, (c & 1) ? 0 : 2, ((c >> 1) & 1) ? 0 : 1, 0, c >= 4 ? 4 : 3};
    return true;
  }
  if (code == 106) { *d = {3, 0, 0, 0, 1}; return true; }  // GRAY_420 (= GRAY_NV21 / NV12 / YV12 / IYUV / I420)
  switch (code) {
    case 107: *d = {2, 2, 0, 1, 3}; return true;   // RGB_UYVY   (U Y0 V Y1)
    case 108: *d = {2, 0, 0, 1, 3}; return true;   // BGR_UYVY
    case 111: *d = {2, 2, 0, 1, 4}; return true;   // RGBA_UYVY
    case 112: *d = {2, 0, 0, 1, 4}; return true;   // BGRA_UYVY
    case 115: *d = {2, 2, 1, 0, 3}; return true;   // RGB_YUY2   (Y0 U Y1 V)
    case 116: *d = {2, 0, 1, 0, 3}; return true;   // BGR_YUY2
    case 117: *d = {2, 2, 3, 0, 3}; return true;   // RGB_YVYU   (Y0 V Y1 U)
    case 118: *d = {2, 0, 3, 0, 3}; return true;   // BGR_YVYU
    case 119: *d = {2, 2, 1, 0, 4}; return true;   // RGBA_YUY2
    case 120: *d = {2, 0, 1, 0, 4}; return true;   // BGRA_YUY2
    case 121: *d = {2, 2, 3, 0, 4}; return true;   // RGBA_YVYU
    case 122: *d = {2, 0, 3, 0, 4}; return true;   // BGRA_YVYU
    case 123: *d = {4, 0, 0, 1, 1}; return true;   // GRAY_UYVY
    case 124: *d = {4, 0, 1, 0, 1}; return true;   // GRAY_YUY2 (= YVYU / YUYV / YUNV)
    default: return false;
  }
}

struct YuvArgsK {
  const uint8_t* const* src;
  uint8_t* const* dst;
  int H, W;   // OUTPUT size
  YuvDesc d;
};

__device__ __forceinline__ void yuv_px(int Y, int u, int v, int d[3]) {  // d = (b, g, r)
  const int yy = max(0, Y - 16) * 1220542;
  d[2] = cvt_sat((yy + (1 << 19) + 1673527 * v) >> 20);
  d[1] = cvt_sat((yy + (1 << 19) - 852492 * v - 409993 * u) >> 20);
  d[0] = cvt_sat((yy + (1 << 19) + 2116026 * u) >> 20);
}

// byte-wise front-end: any even size, any alignment
__global__ __launch_bounds__(256) void k_cvt_yuv_u8(YuvArgsK a) {
  const uint8_t* __restrict__ src = st_gl(st_gl(a.src[blockIdx.y]));
  uint8_t* __restrict__ dst = st_gl(st_gl(a.dst[blockIdx.y]));
  const long long npix = (long long)a.H * a.W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
    const int y = (int)(i / a.W), x = (int)(i - (long long)y * a.W);
    int Y, U = 128, V = 128;
    if (a.d.kind == 0 || a.d.kind == 1 || a.d.kind == 3) {
      Y = src[i];
      if (a.d.kind == 0) {
        const uint8_t* uv = src + npix + (size_t)(y >> 1) * a.W + 2 * (x >> 1);
        U = uv[a.d.uidx]; V = uv[1 - a.d.uidx];
      } else if (a.d.kind == 1) {
        const size_t q = (size_t)(a.H >> 1) * (a.W >> 1), o = (size_t)(y >> 1) * (a.W >> 1) + (x >> 1);
        const uint8_t* p0 = src + npix;  // first chroma plane
        U = a.d.uidx == 0 ? p0[o] : p0[q + o];
        V = a.d.uidx == 0 ? p0[q + o] : p0[o];
      }
    } else {
      const uint8_t* g = src + ((size_t)y * a.W + (x & ~1)) * 2;
      Y = g[a.d.yidx + 2 * (x & 1)];
      U = g[a.d.uidx]; V = g[a.d.uidx ^ 2];
    }
    if (a.d.dcn == 1) {
      dst[i] = (uint8_t)Y;
      continue;
    }
    int c[3];
    yuv_px(Y, U - 128, V - 128, c);
    uint8_t* o = dst + i * a.d.dcn;
    o[a.d.bidx] = (uint8_t)c[0];
    o[1] = (uint8_t)c[1];
    o[a.d.bidx ^ 2] = (uint8_t)c[2];
    if (a.d.dcn == 4) o[3] = 255;
  }
}

// PX = 4 or 16 pixels of a row per thread through whole dwords (W a multiple of PX; frames 4- / 16-byte aligned): PX luma
// bytes, the PX / 2 chroma pairs they share (interleaved chroma, or PX / 2 bytes of each plane, or the 4:2:2 groups)
template <int PX>
__device__ __forceinline__ void yuv_load(const uint8_t* p, unsigned* w) {  // PX bytes -> PX / 4 dwords
  if (PX == 16) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
  } else {
    w[0] = *reinterpret_cast<const unsigned*>(p);
  }
}
__device__ __forceinline__ int yuv_byte(const unsigned* w, int k) { return (int)((w[k >> 2] >> (8 * (k & 3))) & 0xffu); }

template <int DCN, int PX>
__global__ __launch_bounds__(256) void k_cvt_yuv_u8_vec(YuvArgsK a) {
  const uint8_t* __restrict__ src = st_gl(st_gl(a.src[blockIdx.y]));
  unsigned* __restrict__ dst = reinterpret_cast<unsigned*>(st_gl(a.dst[blockIdx.y]));
  const long long npix = (long long)a.H * a.W, groups = npix / PX;
  const int gw = a.W / PX;  // groups per row
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < groups; i += (long long)gridDim.x * 256) {
    const int y = (int)(i / gw), xg = (int)(i - (long long)y * gw);
    int Y[PX], U[PX / 2], V[PX / 2];
    if (a.d.kind == 2 || a.d.kind == 4) {
      unsigned w[PX / 2];  // 2 bytes per pixel
      yuv_load<PX>(src + i * 2 * PX, w);
      yuv_load<PX>(src + i * 2 * PX + PX, w + PX / 4);
#pragma unroll
      for (int q = 0; q < PX / 2; ++q) {
        Y[2 * q] = yuv_byte(w, 4 * q + a.d.yidx); Y[2 * q + 1] = yuv_byte(w, 4 * q + a.d.yidx + 2);
        U[q] = yuv_byte(w, 4 * q + a.d.uidx); V[q] = yuv_byte(w, 4 * q + (a.d.uidx ^ 2));
      }
    } else {
      unsigned yw[PX / 4];
      yuv_load<PX>(src + i * PX, yw);
#pragma unroll
      for (int k = 0; k < PX; ++k) Y[k] = yuv_byte(yw, k);
#pragma unroll
      for (int q = 0; q < PX / 2; ++q) U[q] = V[q] = 128;
      if (a.d.kind == 0) {
        unsigned cw[PX / 4];
        yuv_load<PX>(src + npix + (size_t)(y >> 1) * a.W + (size_t)PX * xg, cw);
#pragma unroll
        for (int q = 0; q < PX / 2; ++q) {
          const int c0 = yuv_byte(cw, 2 * q), c1 = yuv_byte(cw, 2 * q + 1);
          U[q] = a.d.uidx ? c1 : c0; V[q] = a.d.uidx ? c0 : c1;
        }
      } else if (a.d.kind == 1) {
        const size_t q = (size_t)(a.H >> 1) * (a.W >> 1), o = (size_t)(y >> 1) * (a.W >> 1) + (size_t)(PX / 2) * xg;
        unsigned p0[2], p1[2];  // PX / 2 bytes of each plane: 2 (PX = 4) or 8 (PX = 16)
        if (PX == 16) {
          const uint2 v0 = *reinterpret_cast<const uint2*>(src + npix + o), v1 = *reinterpret_cast<const uint2*>(src + npix + q + o);
          p0[0] = v0.x; p0[1] = v0.y; p1[0] = v1.x; p1[1] = v1.y;
        } else {
          p0[0] = *reinterpret_cast<const unsigned short*>(src + npix + o); p1[0] = *reinterpret_cast<const unsigned short*>(src + npix + q + o);
          p0[1] = p1[1] = 0;
        }
#pragma unroll
        for (int k = 0; k < PX / 2; ++k) {
          const int f = yuv_byte(p0, k), g = yuv_byte(p1, k);
          U[k] = a.d.uidx == 0 ? f : g; V[k] = a.d.uidx == 0 ? g : f;
        }
      }
    }
    constexpr int NO = DCN * PX / 4;
    unsigned out[NO];
#pragma unroll
    for (int k = 0; k < NO; ++k) out[k] = 0;
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      int d[4];
      if (DCN == 1) {
        d[0] = Y[p];
      } else {
        int c[3];
        yuv_px(Y[p], U[p >> 1] - 128, V[p >> 1] - 128, c);
        d[0] = a.d.bidx ? c[2] : c[0]; d[1] = c[1]; d[2] = a.d.bidx ? c[0] : c[2]; d[3] = 255;
      }
#pragma unroll
      for (int k = 0; k < DCN; ++k) out[(p * DCN + k) >> 2] |= ((unsigned)d[k] & 0xffu) << (8 * ((p * DCN + k) & 3));
    }
    if (PX == 16) {
#pragma unroll
      for (int k = 0; k < NO / 4; ++k)
        reinterpret_cast<uint4*>(dst + i * NO)[k] = make_uint4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
    } else {
#pragma unroll
      for (int k = 0; k < NO; ++k) dst[i * NO + k] = out[k];
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
  void (*kern)(BlurArgsK) = kernel_size == 3 ? k_box_blur_u8c3<3> : kernel_size == 5 ? k_box_blur_u8c3<5>
                          : kernel_size == 7 ? k_box_blur_u8c3<7> : kernel_size == 9 ? k_box_blur_u8c3<9>
                          : kernel_size == 11 ? k_box_blur_u8c3<11> : k_box_blur_u8c3<0>;
  // (9 and 11: 617 -> 399 and 757 -> 571 us per 64 1080p frames against the generic path; from 13 on the static register ring
  // of the specialisation no longer fits -- 310+ registers, one wave per SIMD -- and the generic path is faster)
  ST_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nb = 3 * w;
  for (int f0 = 0; f0 < n; f0 += 65535) {
    const int nf = n - f0 < 65535 ? n - f0 : 65535;
    a.src = d_src + f0; a.dst = d_dst + f0;
    dim3 grid((nb + BL_TILEB - 1) / BL_TILEB, (h + BL_ROWS - 1) / BL_ROWS, nf);
    if (grid.y > 65535) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "blur: frame too tall");
    st_timed t(ctx, ST_K_BLUR_OP);
    hipLaunchKernelGGL(kern, grid, dim3(BL_T), lds, ctx->stream, a);
    ST_HIP(ctx, hipGetLastError());
  }
  return ST_OK;
}

namespace {
// cv::interpolateLanczos4 (imgproc/src/resize.cpp)
void lanczos4_coeffs(float x, float* coeffs) {
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
  if (x < FLT_EPSILON) {
    for (int i = 0; i < 8; i++) coeffs[i] = 0;
    coeffs[3] = 1;
    return;
  }
  float sum = 0;
  const double y0 = -(x + 3) * 3.1415926535897932384626433832795 * 0.25, s0 = std::sin(y0), c0 = std::cos(y0);
  for (int i = 0; i < 8; i++) {
    const double y = -(x + 3 - i) * 3.1415926535897932384626433832795 * 0.25;
    coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
    sum += coeffs[i];
  }
  sum = 1.f / sum;
  for (int i = 0; i < 8; i++) coeffs[i] *= sum;
}
void lanczos4_axis(int dsize, double scale, int* ofs, short* coef) {
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    const int sidx = (int)floorf(f);
    f -= sidx;
    float c[8];
    lanczos4_coeffs(f, c);
    ofs[d] = sidx;
    for (int k = 0; k < 8; ++k) {
      const long r = lrintf(c[k] * 2048);  // saturate_cast<short>: round half to even
      coef[8 * d + k] = (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
    }
  }
}
}  // namespace

ST_EXPORT int st_resize_u8_batch(st_ctx* ctx, const uint8_t* const* frames_dev, int n, int h, int w, int channels,
                                 int out_h, int out_w, int interpolation, uint8_t* const* out_dev) {
  ST_TRY(st_enter(ctx));
  if (n < 0 || h <= 0 || w <= 0 || out_h <= 0 || out_w <= 0 || channels < 1 || channels > 4 ||
      (long long)h * w > 200000000LL || (long long)out_h * out_w > 200000000LL)
    return st_set_error(ctx, ST_ERR_INVALID, "resize: bad arguments (n=%d %dx%dx%d -> %dx%d)", n, h, w, channels, out_h, out_w);
  if (interpolation < ST_INTER_NEAREST || interpolation > ST_INTER_LANCZOS4)
    return st_set_error(ctx, ST_ERR_UNSUPPORTED, "resize: interpolation %d (INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA and INTER_LANCZOS4 are implemented)", interpolation);
  if (out_h > 65535) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "resize: output taller than 65535 rows");
  if (n == 0) return ST_OK;
  if (!frames_dev || !out_dev) return st_set_error(ctx, ST_ERR_INVALID, "resize: null argument");
  for (int i = 0; i < n; ++i)
    if (!frames_dev[i] || !out_dev[i] || frames_dev[i] == out_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "resize: row %d is null or aliased", i);
  const size_t tb = st_align_up(sizeof(void*) * (size_t)n);
  const bool lanczos = interpolation == ST_INTER_LANCZOS4 && !(h == out_h && w == out_w);
  const size_t lz = lanczos ? st_align_up(sizeof(int) * (size_t)out_w) + st_align_up(sizeof(short) * 8 * (size_t)out_w) +
                              st_align_up(sizeof(int) * (size_t)out_h) + st_align_up(sizeof(short) * 8 * (size_t)out_h) : 0;
  ST_TRY(st_ws_reserve(ctx, 2 * tb + lz));
  const uint8_t** d_src = (const uint8_t**)st_ws_alloc(ctx, tb);
  uint8_t** d_dst = (uint8_t**)st_ws_alloc(ctx, tb);
  ST_HIP(ctx, hipMemcpyAsync(d_src, frames_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  ST_HIP(ctx, hipMemcpyAsync(d_dst, out_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  ResizeArgsK a;
  memset(&a, 0, sizeof(a));
  a.sh = h; a.sw = w; a.dh = out_h; a.dw = out_w; a.cn = channels;
  const double inv_sx = (double)out_w / w, inv_sy = (double)out_h / h;
  a.scale_x = 1. / inv_sx; a.scale_y = 1. / inv_sy;
  a.inv_scale_x = inv_sx; a.inv_scale_y = inv_sy;
  a.iscale_x = (int)lrint(a.scale_x); a.iscale_y = (int)lrint(a.scale_y);
  const bool area_fast = fabs(a.scale_x - a.iscale_x) < DBL_EPSILON && fabs(a.scale_y - a.iscale_y) < DBL_EPSILON;
  if (h == out_h && w == out_w) a.mode = RS_COPY;
  else if (interpolation == ST_INTER_NEAREST) a.mode = RS_NEAREST;
  else if (interpolation == ST_INTER_CUBIC) a.mode = RS_CUBIC;
  else if (interpolation == ST_INTER_LANCZOS4) a.mode = RS_LANCZOS4;
  else if ((interpolation == ST_INTER_LINEAR || interpolation == ST_INTER_AREA) && area_fast && a.iscale_x == 2 && a.iscale_y == 2) a.mode = RS_AREA2;
  else if (interpolation == ST_INTER_LINEAR) a.mode = RS_LINEAR;
  else if (a.scale_x >= 1 && a.scale_y >= 1) a.mode = area_fast ? RS_AREA_INT : RS_AREA;
  else a.mode = RS_LINEAR_AREA;
  std::vector<int> hx, hy;
  std::vector<short> ha, hb;
  if (a.mode == RS_LANCZOS4) {
    hx.resize(out_w); ha.resize(8 * (size_t)out_w); hy.resize(out_h); hb.resize(8 * (size_t)out_h);
    lanczos4_axis(out_w, a.scale_x, hx.data(), ha.data());
    lanczos4_axis(out_h, a.scale_y, hy.data(), hb.data());
    int* dx_ = (int*)st_ws_alloc(ctx, sizeof(int) * (size_t)out_w);
    short* da_ = (short*)st_ws_alloc(ctx, sizeof(short) * 8 * (size_t)out_w);
    int* dy_ = (int*)st_ws_alloc(ctx, sizeof(int) * (size_t)out_h);
    short* db_ = (short*)st_ws_alloc(ctx, sizeof(short) * 8 * (size_t)out_h);
    if (!dx_ || !da_ || !dy_ || !db_) return st_set_error(ctx, ST_ERR_OOM, "resize: scratch plan exhausted");
    // pageable sources: hipMemcpyAsync returns once the runtime has staged them, so the vectors may go
    ST_HIP(ctx, hipMemcpyAsync(dx_, hx.data(), sizeof(int) * hx.size(), hipMemcpyHostToDevice, ctx->stream));
    ST_HIP(ctx, hipMemcpyAsync(da_, ha.data(), sizeof(short) * ha.size(), hipMemcpyHostToDevice, ctx->stream));
    ST_HIP(ctx, hipMemcpyAsync(dy_, hy.data(), sizeof(int) * hy.size(), hipMemcpyHostToDevice, ctx->stream));
    ST_HIP(ctx, hipMemcpyAsync(db_, hb.data(), sizeof(short) * hb.size(), hipMemcpyHostToDevice, ctx->stream));
    a.xofs = dx_; a.ialpha = da_; a.yofs = dy_; a.ibeta = db_;
  }
  for (int f0 = 0; f0 < n; f0 += 65535) {
    const int nf = n - f0 < 65535 ? n - f0 : 65535;
    a.src = d_src + f0; a.dst = d_dst + f0;
    st_timed t(ctx, ST_K_RESIZE);
    const bool fast = a.mode == RS_LINEAR && channels == 3;
    if (a.mode == RS_AREA2 && channels == 3)
      hipLaunchKernelGGL(k_resize_area2_c3_v4, dim3(((out_w + 3) / 4 + 255) / 256, out_h, nf), dim3(256), 0, ctx->stream, a);
    else if (a.mode == RS_AREA && channels == 3 && a.scale_x <= 6) {  // a cell spans at most scale + 2 source columns
      const dim3 gr(((out_w + 3) / 4 + 255) / 256, (out_h + RL_ROWS - 1) / RL_ROWS, nf);
      if (a.scale_x <= 2) hipLaunchKernelGGL(k_resize_area_c3_v4<4>, gr, dim3(256), 0, ctx->stream, a);
      else hipLaunchKernelGGL(k_resize_area_c3_v4<8>, gr, dim3(256), 0, ctx->stream, a);
    }
    else if (a.mode == RS_AREA_INT && channels == 3 && a.iscale_x >= 2 && a.iscale_x <= 4) {
      const dim3 gr(((out_w + 3) / 4 + 255) / 256, out_h, nf);
      if (a.iscale_x == 2) hipLaunchKernelGGL(k_resize_area_int_c3_v4<2>, gr, dim3(256), 0, ctx->stream, a);
      else if (a.iscale_x == 3) hipLaunchKernelGGL(k_resize_area_int_c3_v4<3>, gr, dim3(256), 0, ctx->stream, a);
      else hipLaunchKernelGGL(k_resize_area_int_c3_v4<4>, gr, dim3(256), 0, ctx->stream, a);
    }
    else if (a.mode == RS_LANCZOS4 && channels == 3)
      hipLaunchKernelGGL(k_resize_lanczos4_c3, dim3((out_w + 255) / 256, out_h, nf), dim3(256), 0, ctx->stream, a);
    else if (a.mode == RS_NEAREST && channels == 3)
      hipLaunchKernelGGL(k_resize_nearest_c3_v4, dim3(((out_w + 3) / 4 + 255) / 256, (out_h + RL_ROWS - 1) / RL_ROWS, nf), dim3(256), 0, ctx->stream, a);
    else if (a.mode == RS_CUBIC && channels == 3)
      hipLaunchKernelGGL(k_resize_cubic_c3_v4, dim3(((out_w + 3) / 4 + 255) / 256, (out_h + RL_ROWS - 1) / RL_ROWS, nf), dim3(256), 0, ctx->stream, a);
    else if (a.mode == RS_LINEAR_AREA && channels == 3)
      hipLaunchKernelGGL(k_resize_linear_c3_v4<true>, dim3(((out_w + 3) / 4 + 255) / 256, (out_h + RL_ROWS - 1) / RL_ROWS, nf), dim3(256), 0, ctx->stream, a);
    else if (fast)
      hipLaunchKernelGGL(k_resize_linear_c3_v4<false>, dim3(((out_w + 3) / 4 + 255) / 256, (out_h + RL_ROWS - 1) / RL_ROWS, nf), dim3(256), 0, ctx->stream, a);
    else
      hipLaunchKernelGGL(k_resize_u8, dim3((out_w + 255) / 256, out_h, nf), dim3(256), 0, ctx->stream, a);
    ST_HIP(ctx, hipGetLastError());
  }
  return ST_OK;
}

ST_EXPORT int st_cvt_color_out_channels(int code, int in_channels) {
  switch (code) {
    case ST_COLOR_BGR2RGB: case ST_COLOR_BGR2HSV: case ST_COLOR_BGR2YCrCb: case ST_COLOR_RGB2YCrCb:
    case ST_COLOR_YCrCb2BGR: case ST_COLOR_YCrCb2RGB: case ST_COLOR_RGB2HSV: case ST_COLOR_HSV2BGR: case ST_COLOR_HSV2RGB:
    case ST_COLOR_BGR2HSV_FULL: case ST_COLOR_RGB2HSV_FULL: case ST_COLOR_HSV2BGR_FULL: case ST_COLOR_HSV2RGB_FULL:
    case ST_COLOR_BGR2HLS: case ST_COLOR_RGB2HLS: case ST_COLOR_HLS2BGR: case ST_COLOR_HLS2RGB:
    case ST_COLOR_BGR2HLS_FULL: case ST_COLOR_RGB2HLS_FULL: case ST_COLOR_HLS2BGR_FULL: case ST_COLOR_HLS2RGB_FULL:
    case ST_COLOR_BGR2YUV: case ST_COLOR_RGB2YUV: case ST_COLOR_YUV2BGR: case ST_COLOR_YUV2RGB:
    case ST_COLOR_BGR2XYZ: case ST_COLOR_RGB2XYZ: case ST_COLOR_XYZ2BGR: case ST_COLOR_XYZ2RGB:
      return in_channels == 3 ? 3 : -1;
    case ST_COLOR_BGR2GRAY: case ST_COLOR_RGB2GRAY: return in_channels == 3 ? 1 : -1;
    case ST_COLOR_GRAY2BGR: return in_channels == 1 ? 3 : -1;
    default: break;
  }
  CvtLayout l;
  if (cvt_layout_of(code, &l)) return in_channels == cvt_kind_channels(l.sk) ? cvt_kind_channels(l.dk) : -1;
  return -1;
}

ST_EXPORT int st_cvt_color_out_shape(int code, int in_h, int in_w, int in_channels, int* out_h, int* out_w, int* out_channels) {
  YuvDesc d;
  int oh = in_h, ow = in_w, oc;
  if (cvt_yuv_of(code, &d)) {
    if (d.kind == 0 || d.kind == 1 || d.kind == 3) {
      // 4:2:0: a (3H/2, W) single-channel frame with even W and H
      if (in_channels != 1 || in_h <= 0 || in_w <= 0 || in_h % 3 || in_w % 2 || (in_h / 3 * 2) % 2) return -1;
      oh = in_h / 3 * 2;
    } else {
      if (in_channels != 2 || in_h <= 0 || in_w <= 0 || in_w % 2) return -1;
    }
    oc = d.dcn;
  } else {
    oc = st_cvt_color_out_channels(code, in_channels);
    if (oc < 0) return -1;
  }
  if (out_h) *out_h = oh;
  if (out_w) *out_w = ow;
  if (out_channels) *out_channels = oc;
  return 0;
}

ST_EXPORT int st_cvt_color_u8_batch(st_ctx* ctx, const uint8_t* const* frames_dev, int n, int h, int w, int channels,
                                    int code, int gray_bits, uint8_t* const* out_dev) {
  ST_TRY(st_enter(ctx));
  if (n < 0 || h <= 0 || w <= 0 || (long long)h * w > 200000000LL)
    return st_set_error(ctx, ST_ERR_INVALID, "cvt_color: bad arguments (n=%d h=%d w=%d)", n, h, w);
  int oh = 0, ow = 0, oc = 0;
  if (st_cvt_color_out_shape(code, h, w, channels, &oh, &ow, &oc) != 0)
    return st_set_error(ctx, ST_ERR_UNSUPPORTED, "cvt_color: conversion code %d on %dx%d frames of %d channel(s) is not implemented", code, w, h, channels);
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
  YuvDesc yd;
  if (cvt_yuv_of(code, &yd)) {
    YuvArgsK ya;
    ya.H = oh; ya.W = ow; ya.d = yd;
    unsigned align = 0;
    for (int i = 0; i < n; ++i) align |= (unsigned)((uintptr_t)frames_dev[i] | (uintptr_t)out_dev[i]);
    // 16 pixels per thread need 16-byte aligned rows of luma AND of chroma (planar chroma rows are W / 2 bytes: W % 32)
    const bool planar = yd.kind == 1;
    const bool packed422 = yd.kind == 2 || yd.kind == 4;  // 80 bytes per thread at 16 pixels: measured slower than 4 (3.3 vs 4.0 TB/s)
    const int px = (!packed422 && ow % (planar ? 32 : 16) == 0 && (align & 15) == 0 && (!planar || ((long long)oh * ow / 4) % 16 == 0)) ? 16
                   : ((ow % 4 == 0 && (align & 3) == 0) ? 4 : 1);
    long long by = (((long long)oh * ow) / px + 255) / 256;
    if (by > 8192) by = 8192;
    for (int f0 = 0; f0 < n; f0 += 65535) {
      const int nf = n - f0 < 65535 ? n - f0 : 65535;
      ya.src = d_src + f0; ya.dst = d_dst + f0;
      st_timed t(ctx, ST_K_CVT_COLOR);
      const dim3 grid((unsigned)by, nf);
#define YUV_VEC(D_)                                                                                       \
  do {                                                                                                    \
    if (px == 16) hipLaunchKernelGGL((k_cvt_yuv_u8_vec<D_, 16>), grid, dim3(256), 0, ctx->stream, ya);    \
    else hipLaunchKernelGGL((k_cvt_yuv_u8_vec<D_, 4>), grid, dim3(256), 0, ctx->stream, ya);               \
  } while (0)
      if (px == 1) hipLaunchKernelGGL(k_cvt_yuv_u8, grid, dim3(256), 0, ctx->stream, ya);
      else if (yd.dcn == 1) YUV_VEC(1);
      else if (yd.dcn == 3) YUV_VEC(3);
      else YUV_VEC(4);
#undef YUV_VEC
      ST_HIP(ctx, hipGetLastError());
    }
    return ST_OK;
  }
  CvtArgsK a;
  a.npix = (long long)h * w; a.code = code;
  if (gray_bits == 14) { a.cb = 1868; a.cg = 9617; a.cr = 4899; } else { a.cb = 3735; a.cg = 19235; a.cr = 9798; }
  a.shift = gray_bits; a.rnd = 1 << (gray_bits - 1);
  a.bi = code == ST_COLOR_RGB2GRAY ? 2 : 0;
  CvtLayout lay;
  a.layout = cvt_layout_of(code, &lay) ? 1 : 0;
  if (a.layout) { a.sk = lay.sk; a.sbi = lay.sbi; a.dk = lay.dk; a.dbi = lay.dbi; } else { a.sk = a.sbi = a.dk = a.dbi = 0; }
  a.scn = channels; a.dcn = oc;
  // whole-dword paths: 16 (4) pixels per thread when every frame starts on a 16- (4-) byte boundary and holds a multiple of
  // 16 (4) pixels
  unsigned align = 0;
  for (int i = 0; i < n; ++i) align |= (unsigned)((uintptr_t)frames_dev[i] | (uintptr_t)out_dev[i]);
  const int px = (a.npix % 16 == 0 && (align & 15) == 0) ? 16 : ((a.npix % 4 == 0 && (align & 3) == 0) ? 4 : 1);
  long long bx = (a.npix / px + 255) / 256;
  if (bx > 8192) bx = 8192;
  for (int f0 = 0; f0 < n; f0 += 65535) {
    const int nf = n - f0 < 65535 ? n - f0 : 65535;
    a.src = d_src + f0; a.dst = d_dst + f0;
    st_timed t(ctx, ST_K_CVT_COLOR);
    const dim3 grid((unsigned)bx, nf);
#define CVT_V4(S_, D_)                                                                                            \
  do {                                                                                                            \
    if (px == 16) hipLaunchKernelGGL((k_cvt_color_u8_vec<S_, D_, 16>), grid, dim3(256), 0, ctx->stream, a);       \
    else hipLaunchKernelGGL((k_cvt_color_u8_vec<S_, D_, 4>), grid, dim3(256), 0, ctx->stream, a);                  \
  } while (0)
    const int key = px > 1 ? 10 * a.scn + a.dcn : 0;
    switch (key) {
      case 33: CVT_V4(3, 3); break;
      case 31: CVT_V4(3, 1); break;
      case 13: CVT_V4(1, 3); break;
      case 34: CVT_V4(3, 4); break;
      case 43: CVT_V4(4, 3); break;
      case 44: CVT_V4(4, 4); break;
      case 14: CVT_V4(1, 4); break;
      case 41: CVT_V4(4, 1); break;
      case 32: CVT_V4(3, 2); break;
      case 23: CVT_V4(2, 3); break;
      case 42: CVT_V4(4, 2); break;
      case 24: CVT_V4(2, 4); break;
      case 12: CVT_V4(1, 2); break;
      case 21: CVT_V4(2, 1); break;
      default: hipLaunchKernelGGL(k_cvt_color_u8, grid, dim3(256), 0, ctx->stream, a); break;
    }
#undef CVT_V4
    ST_HIP(ctx, hipGetLastError());
  }
  return ST_OK;
}
