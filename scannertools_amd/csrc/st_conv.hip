// Convolution stack of the pose network (BASELINE config 5) for gfx950:
// stride-1 "same" convolutions (3x3, 7x7, 1x1) + bias + ReLU as implicit GEMMs on the matrix cores,
// and the 2x2 max pooling between the VGG blocks.  This is the only dense contraction in the
// repository's scope, hence the only MFMA code.
//
// What it replaces: the Caffe forward pass behind the reference's CPM2 op
// (/root/reference/scannertools_caffe/scannertools_caffe_cpp/cpm2_kernel.cpp:8-52 -> CaffeKernel::execute,
// caffe_kernel.cpp) for the layers of the OpenPose COCO model (profiles/NOTES.md, Part II section 9).  The reference
// computes in float32 (Caffe), so this does too: v_mfma_f32_32x32x2_f32, f32 operands, f32
// accumulation -- bit for bit a k-ordered fmaf chain -- at the f32 matrix rate (157 TFLOP/s peak).
//
// Layout: activations NHWC float32 with a channel count padded to a multiple of 16 (pad channels are
// zero); a layer may read / write a channel SLICE of a wider buffer (pixel stride + channel offset),
// which is how the stage inputs concat(PAF, heat maps, features) are formed without a copy.
// Weights [Cout_pad][KH][KW][Cin] (Cout padded to the block's column count with zero rows).
//
// GEMM view: M = N*H*W output pixels, N = Cout, K = KH*KW*Cin walked as (kh, kw, 16-channel slice).
// Block = 128 pixels x BN (128 or 64) output channels, 256 threads = 4 waves; A (pixels x 16 channels
// of the input shifted by (kh, kw), zero outside the image) and B (BN x 16 weights) are staged through
// LDS k-major, so that a wave's 32 lanes read 32 consecutive rows / columns of one k: the operand
// layout of the 32x32x2 instruction (lane l: A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31]).
// Global loads: 4 adjacent lanes fetch the 64 contiguous bytes of one pixel's slice.  The next slice's
// loads are in flight while the current one is multiplied (registers -> other LDS buffer after the
// MFMAs; one barrier per slice), and inside a slice the operands of step k+2 are read from LDS before
// the MFMAs of step k.  A 16-channel slice keeps the kernel at 111 VGPRs / 33 KB of LDS = 4 blocks per
// CU; 32-channel slices (2 blocks per CU) measured 13 % slower (profiles/README.md).
// Round 4: 3x3 / 7x7 layers with whole 128-channel output blocks run on the spatial-tile kernels further down
// (k_conv_tile_f32, k_conv_tile_bf16x3: the tile's input region held in LDS across the taps, weights in operand order from L1);
// the per-tap kernels here keep the 1x1 layers, the 64-channel-output layers and maps no tile shape fits.
#include <cstdlib>
#include <cstring>

#include "st_conv_tile.h"

namespace {


// LDS row padding: with row stride 130 the 4 channel quads x 8 pixels a half-wave stores (LD = 1) fall into 32 different banks
constexpr int CV_BM = 128, CV_PAD = 2;

struct ConvArgs {
  const float* x;     // input activations
  const float* w;     // [cout_pad][kh][kw][cin]
  const float* bias;  // [cout_pad]
  float* y;           // output activations
  int n, h, wd, cin, xs, xoff;  // xs: floats per input pixel (buffer channel count), xoff: first channel read
  int kh, kw, pad;
  int cout, ys, yoff, relu;
  long long m;        // n * h * wd
};

// BN: output channels per block (128 or 64); BK: input channels per K slice;
// LD: 0 = a wave's lanes load 64 consecutive pixels (one channel quad each), 1 = BK/4 adjacent lanes load one
// pixel's whole slice (contiguous bytes); PF: operands of step kk+2 are read from LDS before the MFMAs of kk
template <int BN, int BK, int LD, int PF>
__global__ __launch_bounds__(256, 3) void k_conv_nhwc_f32(ConvArgs a) {
  constexpr int NA4 = CV_BM * BK / 4 / 256;  // float4 loads of the activation tile per thread
  constexpr int NB4 = BN * BK / 4 / 256;     // ... of the weight tile
  constexpr int QP = BK / 4;                 // channel quads per slice
  constexpr int QA = 256 / CV_BM, QB = 256 / BN;  // LD 0: channel quads covered per load round
  constexpr int PR = 256 / QP;               // LD 1: pixels (weight rows) covered per load round
  __shared__ float As[2][BK][CV_BM + CV_PAD];
  __shared__ float Bs[2][BK][BN + CV_PAD];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const long long m0 = (long long)blockIdx.x * CV_BM;
  const int n0 = blockIdx.y * BN;

  // this thread's pixels of the A tile (fixed over the K walk) and its channel quad(s)
  constexpr int NPX = LD ? NA4 : 1;
  int px[NPX], py[NPX];
  bool mvalid[NPX];
  const float* __restrict__ xpix[NPX];
  const int pm0 = LD ? t / QP : t & (CV_BM - 1), cqa = LD ? t % QP : t / CV_BM;
#pragma unroll
  for (int j = 0; j < NPX; ++j) {
    const long long gm = m0 + pm0 + PR * j;
    mvalid[j] = gm < a.m;
    int pn = 0;
    px[j] = py[j] = 0;
    if (mvalid[j]) {
      px[j] = (int)(gm % a.wd);
      const long long r = gm / a.wd;
      py[j] = (int)(r % a.h);
      pn = (int)(r / a.h);
    }
    xpix[j] = a.x + ((size_t)((size_t)pn * a.h + py[j]) * a.wd + px[j]) * a.xs + a.xoff + 4 * cqa;
  }
  // weight tile: output channel(s) and channel quad(s)
  const int nb0 = LD ? t / QP : t & (BN - 1), cqb = LD ? t % QP : t / BN;
  const float* __restrict__ wpix = a.w + (size_t)(n0 + nb0) * a.kh * a.kw * a.cin + 4 * cqb;
  const size_t wrow = (size_t)a.kh * a.kw * a.cin;

  const int cslices = a.cin / BK;
  const int nslices = a.kh * a.kw * cslices;

  // the K walk (channel slice, ky, kx) advances by counters: no division in the loop
  int f_ky = 0, f_kx = 0, f_c = 0;
  float4 ra[NA4], rb[NB4];
  auto fetch = [&]() {
    const long long shift = ((long long)(f_ky - a.pad) * a.wd + (f_kx - a.pad)) * a.xs + f_c * BK;
#pragma unroll
    for (int j = 0; j < NA4; ++j) {
      const int p = LD ? j : 0;
      const int yy = py[p] + f_ky - a.pad, xx = px[p] + f_kx - a.pad;
      const bool inb = mvalid[p] && (unsigned)yy < (unsigned)a.h && (unsigned)xx < (unsigned)a.wd;
      const float* __restrict__ src = xpix[p] + shift + (LD ? 0 : 4 * QA * j);
      ra[j] = inb ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* __restrict__ wsrc = wpix + (size_t)(f_ky * a.kw + f_kx) * a.cin + f_c * BK;
#pragma unroll
    for (int j = 0; j < NB4; ++j) rb[j] = *reinterpret_cast<const float4*>(wsrc + (LD ? PR * j * wrow : (size_t)(4 * QB * j)));
    // slices outer, taps inner: the order the spatial-tile kernel accumulates in, so that the two kernels give the same bits
    // and the launcher may pick either by launch size
    if (++f_kx == a.kw) {
      f_kx = 0;
      if (++f_ky == a.kh) { f_ky = 0; ++f_c; }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NA4; ++j) {
      const int k = LD ? 4 * cqa : 4 * (cqa + QA * j), pm = LD ? pm0 + PR * j : pm0;
      As[buf][k][pm] = ra[j].x; As[buf][k + 1][pm] = ra[j].y; As[buf][k + 2][pm] = ra[j].z; As[buf][k + 3][pm] = ra[j].w;
    }
#pragma unroll
    for (int j = 0; j < NB4; ++j) {
      const int k = LD ? 4 * cqb : 4 * (cqb + QB * j), nb = LD ? nb0 + PR * j : nb0;
      Bs[buf][k][nb] = rb[j].x; Bs[buf][k + 1][nb] = rb[j].y; Bs[buf][k + 2][nb] = rb[j].z; Bs[buf][k + 3][nb] = rb[j].w;
    }
  };

  // wave tile: BN = 128: 2 x 2 waves of 64 x 64; BN = 64: 4 x 1 waves of 32 x 64
  constexpr int MT = BN == 128 ? 2 : 1, NT = 2;
  const int wm = BN == 128 ? (wv >> 1) * 64 : wv * 32;
  const int wn = BN == 128 ? (wv & 1) * 64 : 0;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  fetch();
  stash(0);
  __syncthreads();
  const int l31 = lane & 31, lk = lane >> 5;
  for (int s = 0; s < nslices; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslices) fetch();
    float af[2][MT], bf[2][NT];
    auto rd = [&](int kk, int slot) {
#pragma unroll
      for (int i = 0; i < MT; ++i) af[slot][i] = As[buf][kk + lk][wm + 32 * i + l31];
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[slot][j] = Bs[buf][kk + lk][wn + 32 * j + l31];
    };
    if (PF) rd(0, 0);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int cur = PF ? (kk >> 1) & 1 : 0;
      if (PF) {
        if (kk + 2 < BK) rd(kk + 2, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs that hide them
      } else {
        rd(kk, 0);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
      if (PF) __builtin_amdgcn_sched_barrier(0);
      // the next slice's LDS stores go out half-way through the slice, under this wave's own matrix instructions
      // (+0.5-1 % against storing after the last one)
      if (kk == BK / 2 && s + 1 < nslices) {
        stash(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nn = n0 + wn + 32 * j + l31;
    if (nn >= a.cout) continue;
    const float b = a.bias[nn];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long mm = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (mm < a.m) {
          float v = acc[i][j][r] + b;
          if (a.relu) v = v > 0.f ? v : 0.f;
          a.y[(size_t)mm * a.ys + a.yoff + nn] = v;
        }
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The same convolution on the bf16 matrix pipe with float32-grade accuracy ("bf16x3").
// CDNA4's float32 MFMA rate is 1/16 of its bf16 rate (157 against 2 500 TFLOP/s dense), so the idiomatic way to a
// float32-accurate contraction on this chip is to split every operand into three bf16 terms, v = hi + mid + lo
// (round-to-nearest at each step; 3 x 8 mantissa bits represent a float32 exactly), and to accumulate the six
// products that carry more than 2^-24 of the result -- hi hi, hi mid, mid hi, hi lo, lo hi, mid mid -- in the
// float32 accumulators of v_mfma_f32_32x32x16_bf16.  Every bf16 x bf16 product is exact in float32; what is dropped
// (mid lo, lo mid, lo lo) is below 2^-23 of a product, the size of one float32 rounding.  Six matrix instructions of
// 32 cycles replace eight of 64 (v_mfma_f32_32x32x2_f32 per 16 k): 2.67 x the float32 pipe rate.
// NOT bit-identical to the float32 kernel (different roundings of the same accuracy): opt-in, the default stays the
// float32 instruction, and the tests bound both against torch's float32 convolution.
//
// Operands: activations stay float32 in memory and are split on their way into LDS (44 VALU instructions per thread
// and slice, hidden under the matrix pipe); weights are split once (st_conv_pack_weights_bf16x3) into
// [cout_pad][tap][cin / 16][3 splits][16 channels] bf16, i.e. 96 contiguous bytes per (output channel, K slice).
// LDS: per split a [128 rows][16 k] bf16 tile, 32-byte rows, the two 16-byte halves of a row swapped on rows
// 8..15 mod 16 so that the 16-byte operand reads of 32 consecutive rows touch every bank once.
// Operand layout of the instruction (checked by the known-answer test): lane l holds A[row l & 31][k 8 (l >> 5) .. + 7],
// B[k 8 (l >> 5) .. + 7][col l & 31]; C/D as for the 32x32x2 instruction.
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_pack_weights_bf16x3(const float* __restrict__ w, int cout_pad, int taps, int cin,
                                                             unsigned* __restrict__ out) {
  // thread = (row = cout * taps + tap, channel pair)
  const long long total = (long long)cout_pad * taps * (cin / 2);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cp = (int)(i % (cin / 2));
    const long long row = i / (cin / 2);
    const float* src = w + row * cin + 2 * cp;
    unsigned h, m, l;
    split3(src[0], src[1], h, m, l);
    const int c = 2 * cp, slice = c >> 4, k = c & 15;
    unsigned* dst = out + (((size_t)row * (cin / 16) + slice) * 3) * 8 + (k >> 1);  // 8 dwords (16 bf16) per split
    dst[0] = h; dst[8] = m; dst[16] = l;
  }
}

template <int BN>
__global__ __launch_bounds__(256, 3) void k_conv_nhwc_bf16x3(ConvArgs a, const unsigned* __restrict__ w3) {
  constexpr int BK = 16;
  // [buffer][split][row][8 dwords]
  __shared__ unsigned As[2][3][CV_BM][8];
  __shared__ unsigned Bs[2][3][BN][8];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const long long m0 = (long long)blockIdx.x * CV_BM;
  const int n0 = blockIdx.y * BN;

  // A: thread loads channels 4 cq .. 4 cq + 3 of pixels pm0 and pm0 + 64
  const int pm0 = t >> 2, cq = t & 3;
  int px[2], py[2];
  bool mvalid[2];
  const float* __restrict__ xpix[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long long gm = m0 + pm0 + 64 * j;
    mvalid[j] = gm < a.m;
    int pn = 0;
    px[j] = py[j] = 0;
    if (mvalid[j]) {
      px[j] = (int)(gm % a.wd);
      const long long r = gm / a.wd;
      py[j] = (int)(r % a.h);
      pn = (int)(r / a.h);
    }
    xpix[j] = a.x + ((size_t)((size_t)pn * a.h + py[j]) * a.wd + px[j]) * a.xs + a.xoff + 4 * cq;
  }
  // B: thread loads the 16-byte half `bh` of the three splits of output channel(s) nb0 (+ 128 t-rows for BN = 128: one)
  // (BN = 64: threads 128..255 repeat the loads of 0..127 and store nothing)
  const int nb0 = (t >> 1) & (BN - 1), bh = t & 1;
  const bool bload = (t >> 1) < BN;
  const int cslices = a.cin / BK, taps = a.kh * a.kw;
  const int nslices = taps * cslices;
  const unsigned* __restrict__ wrow = w3 + ((size_t)(n0 + nb0) * taps * cslices) * 24 + 4 * bh;

  int f_ky = 0, f_kx = 0, f_c = 0;
  float4 ra[2];
  uint4 rb0, rb1, rb2;
  auto fetch = [&]() {
    const long long shift = ((long long)(f_ky - a.pad) * a.wd + (f_kx - a.pad)) * a.xs + f_c * BK;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int yy = py[j] + f_ky - a.pad, xx = px[j] + f_kx - a.pad;
      const bool inb = mvalid[j] && (unsigned)yy < (unsigned)a.h && (unsigned)xx < (unsigned)a.wd;
      ra[j] = inb ? *reinterpret_cast<const float4*>(xpix[j] + shift) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const unsigned* __restrict__ ws = wrow + (size_t)((f_ky * a.kw + f_kx) * cslices + f_c) * 24;
    rb0 = *reinterpret_cast<const uint4*>(ws);
    rb1 = *reinterpret_cast<const uint4*>(ws + 8);
    rb2 = *reinterpret_cast<const uint4*>(ws + 16);
    // slices outer, taps inner, as in k_conv_tile_bf16x3: same products in the same order, same bits
    if (++f_kx == a.kw) {
      f_kx = 0;
      if (++f_ky == a.kh) { f_ky = 0; ++f_c; }
    }
  };
  // row r, 16-byte half c -> dword offset inside the [row][8] tile
  auto swz = [](int r, int c) { return r * 8 + ((c ^ ((r >> 3) & 1)) << 2); };
  // the next slice goes to the other LDS buffer in three pieces (activation pixel 0, pixel 1, weights) that the main
  // loop places BETWEEN its matrix instructions: the split arithmetic and the LDS stores of a wave then run while its
  // own MFMAs occupy the matrix pipe
  auto stashA = [&](int buf, int j) {
    unsigned h0, m0_, l0, h1, m1, l1;
    split3(ra[j].x, ra[j].y, h0, m0_, l0);
    split3(ra[j].z, ra[j].w, h1, m1, l1);
    const int r = pm0 + 64 * j;
    const int o = swz(r, cq >> 1) + 2 * (cq & 1);
    *reinterpret_cast<uint2*>(&As[buf][0][0][0] + o) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(&As[buf][1][0][0] + o) = make_uint2(m0_, m1);
    *reinterpret_cast<uint2*>(&As[buf][2][0][0] + o) = make_uint2(l0, l1);
  };
  auto stashB = [&](int buf) {
    if (bload) {
      const int o = swz(nb0, bh);
      *reinterpret_cast<uint4*>(&Bs[buf][0][0][0] + o) = rb0;
      *reinterpret_cast<uint4*>(&Bs[buf][1][0][0] + o) = rb1;
      *reinterpret_cast<uint4*>(&Bs[buf][2][0][0] + o) = rb2;
    }
  };
  auto stash = [&](int buf) { stashA(buf, 0); stashA(buf, 1); stashB(buf); };

  constexpr int MT = BN == 128 ? 2 : 1, NT = 2;
  const int wm = BN == 128 ? (wv >> 1) * 64 : wv * 32;
  const int wn = BN == 128 ? (wv & 1) * 64 : 0;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  fetch();
  stash(0);
  __syncthreads();
  const int l31 = lane & 31, lk = lane >> 5;
  for (int s = 0; s < nslices; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslices) fetch();
    bf16x8 af[3][MT], bfr[3][NT];
    auto rdA = [&](int sp) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
        af[sp][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&As[buf][sp][0][0] + swz(wm + 32 * i + l31, lk)));
    };
    auto rdB = [&](int sp) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
        bfr[sp][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&Bs[buf][sp][0][0] + swz(wn + 32 * j + l31, lk)));
    };
    // one product term over the wave's 2 x 2 tiles: four independent accumulators between two uses of the same one
    auto term = [&](int sa, int sb) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[sa][i], bfr[sb][j], acc[i][j], 0, 0, 0);
    };
    // operands are requested in the order the terms consume them (LDS returns in order: each term waits only for its own)
    rdA(0); rdB(0); rdB(1); rdA(1); rdB(2); rdA(2);
    const bool more = s + 1 < nslices;   // uniform
    term(0, 0);
    term(0, 1);
    term(1, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) stashB(buf ^ 1);
    term(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (more) stashA(buf ^ 1, 0);
    term(0, 2);
    __builtin_amdgcn_sched_barrier(0);
    if (more) stashA(buf ^ 1, 1);
    term(2, 0);
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nn = n0 + wn + 32 * j + l31;
    if (nn >= a.cout) continue;
    const float b = a.bias[nn];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long mm = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (mm < a.m) {
          float v = acc[i][j][r] + b;
          if (a.relu) v = v > 0.f ? v : 0.f;
          a.y[(size_t)mm * a.ys + a.yoff + nn] = v;
        }
      }
  }
}

// (the spatial-tile kernels: st_conv_tile_bf16x3.hip, st_conv_tile_f32.hip; shared declarations: st_conv_tile.h)

// the tile shape for an h x w map and a K x K kernel: the (TH, TW) with the fewest maxpx-pixel instruction blocks per image
// whose region fits the LDS buffer (rpmax pixels); false when even the best wastes more than a quarter of the matrix work
bool conv_tile_plan(int h, int w, int ks, int maxpx, int rpmax, int* th, int* tw, double* eff) {
  double best = 0;
  for (int nx = 1; nx <= w; ++nx) {
    const int cw = (w + nx - 1) / nx;
    if (cw > maxpx) continue;
    if (cw < 8 && nx > 1) break;
    int ch = maxpx / cw;
    if (ch > h) ch = h;
    while (ch > 1 && (ch + ks - 1) * (cw + ks - 1) > rpmax) --ch;
    if ((ch + ks - 1) * (cw + ks - 1) > rpmax) continue;
    const long long tiles = (long long)nx * ((h + ch - 1) / ch);
    const double e = (double)h * w / ((double)maxpx * (double)tiles);
    if (e > best + 1e-9) { best = e; *th = ch; *tw = cw; }
  }
  *eff = best;
  return best >= 0.75;
}

bool conv_tile_weights(int kh, int kw, int cout_pad, int cin) { return kh == kw && (kh == 3 || kh == 7) && cout_pad % 128 == 0 && cin % 16 == 0; }

// Which kernel a 3x3 / 7x7 layer with whole 128-channel output blocks runs on, by launch size: 8 = the 8-wave tile kernel (256-pixel
// tiles, one workgroup per CU; float32 only, `allow8`: there it is 2.7 % ahead over the network at 32 frames per call, in
// bf16x3 the 4-wave instance is level or ahead at every size) when its launch fills three quarters of the chip; else 4 = the
// 4-wave instance (128-pixel tiles, two per CU: 46x82, 7x7, 128 -> 128, 5 frames: bf16x3 0.20 against 0.36 ms (per-tap kernel
// 0.44), float32 0.42 against 0.77 (0.62)); else, when even that leaves CUs idle, 41 = four waves with ONE 32-pixel instruction
// tile each (64-pixel tiles: half the serial work per wave, twice the workgroups; 1 frame: bf16x3 0.10 against 0.17 ms);
// 0 = the per-tap kernel (no tile shape fits).  Fills the geometry of `ta`.  Every choice accumulates an output in the same
// order, so none changes a bit; ST_CONV_TILE=0 forces the per-tap kernel, 1 / 4 / 41 the instance of that name wherever a
// tile shape exists.
int conv_tile_choose(st_ctx* ctx, int n, int h, int w, int kh, int kw, int cout_pad, int cin, bool allow8, ConvTileArgs* ta) {
  if (ctx->conv_tile == 0 || !conv_tile_weights(kh, kw, cout_pad, cin)) return 0;
  int th8 = 0, tw8 = 0, th4 = 0, tw4 = 0, th2 = 0, tw2 = 0;
  double e8 = 0, e4 = 0, e2 = 0;
  const bool ok8 = allow8 && conv_tile_plan(h, w, kh, 256, CT_RPMAX, &th8, &tw8, &e8);
  const bool ok4 = conv_tile_plan(h, w, kh, 128, CT_RPMAX / 2, &th4, &tw4, &e4);
  const bool ok2 = conv_tile_plan(h, w, kh, 64, CT_RPMAX / 2, &th2, &tw2, &e2);   // 4 waves x one 32-pixel instruction tile
  auto wgs = [&](int th, int tw) { return (long long)n * ((w + tw - 1) / tw) * ((h + th - 1) / th) * (cout_pad / 128); };
  // below this many 4-wave workgroups per 100 CUs a wave takes one 32-pixel instruction tile instead of two (measured over the
  // network, frames per call 1 / 2 / 5 / 8: bf16x3 129 / 230 / 304 / 400 frames/s without, 162 / 279 / 332 / 399 with; float32
  // 68 / 126 / 161 / 236 and 102 / 183 / 184 / 235); ST_CONV_MT1_PCT moves the line (experiments)
  static const int mt1_pct = getenv("ST_CONV_MT1_PCT") ? atoi(getenv("ST_CONV_MT1_PCT")) : 150;
  int nw = 0;
  if (ctx->conv_tile == 1) nw = ok8 ? 8 : (ok4 ? 4 : 0);
  else if (ctx->conv_tile == 4) nw = ok4 ? 4 : (ok8 ? 8 : 0);
  else if (ctx->conv_tile == 41) nw = ok2 ? 41 : (ok4 ? 4 : 0);
  else if (ok8 && wgs(th8, tw8) >= (long long)ctx->num_cus * 3 / 4) nw = 8;
  else if (ok4 && (!ok2 || wgs(th4, tw4) * 100 >= (long long)ctx->num_cus * mt1_pct)) nw = 4;
  else if (ok2) nw = 41;
  else if (ok4) nw = 4;
  else if (ok8) nw = 8;
  if (!nw) return 0;
  const int th = nw == 8 ? th8 : (nw == 4 ? th4 : th2), tw = nw == 8 ? tw8 : (nw == 4 ? tw4 : tw2);
  ta->n = n; ta->h = h; ta->wd = w; ta->cin = cin; ta->pad = kh / 2;
  ta->th = th; ta->tw = tw; ta->rw = tw + kh - 1; ta->rp = (th + kh - 1) * (tw + kh - 1);
  ta->tiles_x = (w + tw - 1) / tw; ta->tiles_y = (h + th - 1) / th;
  return nw;
}

struct PoolArgs {
  const float* x;
  float* y;
  int n, h, wd, c, xs, ys;  // output (h/2, wd/2)
};

// 2x2 max pooling, stride 2 (the VGG trunk's pool1..pool3), NHWC, 4 channels per thread
__global__ __launch_bounds__(256) void k_maxpool2_nhwc_f32(PoolArgs a) {
  const int oh = a.h / 2, ow = a.wd / 2, c4 = a.c / 4;
  const long long total = (long long)a.n * oh * ow * c4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int q = (int)(i % c4);
    long long r = i / c4;
    const int ox = (int)(r % ow);
    r /= ow;
    const int oy = (int)(r % oh), n = (int)(r / oh);
    const float* __restrict__ s = a.x + ((size_t)((size_t)n * a.h + 2 * oy) * a.wd + 2 * ox) * a.xs + 4 * q;
    const float4 v00 = *reinterpret_cast<const float4*>(s), v01 = *reinterpret_cast<const float4*>(s + a.xs);
    const float4 v10 = *reinterpret_cast<const float4*>(s + (size_t)a.wd * a.xs),
                 v11 = *reinterpret_cast<const float4*>(s + (size_t)a.wd * a.xs + a.xs);
    float4 o;
    o.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x));
    o.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
    o.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z));
    o.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
    *reinterpret_cast<float4*>(a.y + ((size_t)((size_t)n * oh + oy) * ow + ox) * a.ys + 4 * q) = o;
  }
}

struct PlanarArgs {
  const float* x;  // n x (c, h, w) planar
  float* y;        // n x (h, w, cs) NHWC, channels c .. cs-1 zero
  int n, c, h, wd, cs;
};

// planar (CPM2Input's output) -> NHWC with zero pad channels (the first layer's operand); thread = (pixel, channel
// quad), so that the lanes of a wave store consecutive 16-byte pieces
__global__ __launch_bounds__(256) void k_planar_to_nhwc(PlanarArgs a) {
  const int qpp = a.cs / 4;
  const long long hw = (long long)a.h * a.wd, total = (long long)a.n * hw * qpp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int q = (int)(i % qpp);
    const long long px = i / qpp;
    const int n = (int)(px / hw);
    const long long p = px - (long long)n * hw;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = 4 * q + k;
      v[k] = c < a.c ? a.x[((size_t)n * a.c + c) * hw + p] : 0.f;
    }
    *reinterpret_cast<float4*>(a.y + (size_t)px * a.cs + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

}  // namespace

namespace {
// One or two convolutions of the same geometry (`nops`): checks, kernel choice, launch.  Two go into ONE launch where the
// spatial-tile kernel runs (grid.z), else into one per-tap launch each.  f32: operands' `w` is the float32 tensor and
// `w_tile` its tile-order copy or null; bf16x3: `w` is the packed buffer of st_conv_pack_weights_bf16x3.
int conv_launch(st_ctx* ctx, bool f32, int n, int h, int w, int cin, int kh, int kw, int cout_pad, int relu, const st_conv_operands* ops,
                int nops) {
  if (!ops || n <= 0 || h <= 0 || w <= 0 || cin <= 0) return st_set_error(ctx, ST_ERR_INVALID, "conv2d: bad arguments");
  if (kh != kw || !(kh & 1) || kh > 7) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "conv2d: %dx%d kernels (odd square kernels up to 7 are implemented)", kh, kw);
  if (cout_pad <= 0 || cout_pad % 64) return st_set_error(ctx, ST_ERR_INVALID, "conv2d: cout_pad must be a multiple of 64 >= cout");
  bool tile_weights = true;
  for (int k = 0; k < nops; ++k) {
    const st_conv_operands& o = ops[k];
    if (!o.x || !o.w || !o.bias || !o.y || o.cout <= 0) return st_set_error(ctx, ST_ERR_INVALID, "conv2d: bad arguments");
    if (cin % 16 || o.x_offset % 4 || o.x_stride % 4 || o.x_offset + cin > o.x_stride || ((uintptr_t)o.x & 15) || ((uintptr_t)o.w & 15))
      return st_set_error(ctx, ST_ERR_INVALID, "conv2d: input channels must be a multiple of 16 inside a 16-byte aligned buffer (cin=%d stride=%d offset=%d)", cin, o.x_stride, o.x_offset);
    if (cout_pad < o.cout) return st_set_error(ctx, ST_ERR_INVALID, "conv2d: cout_pad must be a multiple of 64 >= cout");
    if (o.y_offset + o.cout > o.y_stride) return st_set_error(ctx, ST_ERR_INVALID, "conv2d: output slice exceeds the buffer's channel count");
    if (f32 && o.w_tile && ((uintptr_t)o.w_tile & 15)) return st_set_error(ctx, ST_ERR_INVALID, "conv2d: the tile-order weights must be 16-byte aligned");
    if (f32 && !o.w_tile) tile_weights = false;
  }
  const int bn = cout_pad % 128 == 0 ? 128 : 64;
  const long long m = (long long)n * h * w;
  const long long bm = (m + CV_BM - 1) / CV_BM;
  if (bm > 2147483647LL) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "conv2d: too many output pixels");
  // Kernel by launch size (conv_tile_choose); every choice accumulates each output in the same order, so it never changes a bit.
  ConvTileArgs ta;
  const int nw = tile_weights ? conv_tile_choose(ctx, n * nops, h, w, kh, kw, cout_pad, cin, f32, &ta) : 0;
  if (nw) {
    ta.n = n;
    ta.relu = relu ? 1 : 0;
    for (int k = 0; k < nops; ++k) {
      const st_conv_operands& o = ops[k];
      ConvTileOperands& t = ta.op[k];
      t.x = o.x; t.bias = o.bias; t.y = o.y;
      t.w3t = f32 ? (const unsigned*)o.w_tile : (const unsigned*)o.w + (size_t)cout_pad * kh * kw * (cin / 2) * 3;
      t.xs = o.x_stride; t.xoff = o.x_offset; t.cout = o.cout; t.ys = o.y_stride; t.yoff = o.y_offset;
    }
    if (nops == 1) ta.op[1] = ta.op[0];
    const long long tiles = (long long)n * ta.tiles_x * ta.tiles_y;
    if (tiles > 2147483647LL) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "conv2d: too many output pixels");
    dim3 tgrid((unsigned)tiles, cout_pad / 128, nops);
    st_timed t(ctx, ST_K_CONV);
    if (f32) st_conv_tile_launch_f32(ctx, kh, nw, tgrid, ta);
    else st_conv_tile_launch_bf16x3(ctx, kh, nw, tgrid, ta);
    ST_HIP(ctx, hipGetLastError());
    return ST_OK;
  }
  for (int k = 0; k < nops; ++k) {
    const st_conv_operands& o = ops[k];
    ConvArgs a;
    a.x = o.x; a.w = f32 ? (const float*)o.w : nullptr; a.bias = o.bias; a.y = o.y;
    a.n = n; a.h = h; a.wd = w; a.cin = cin; a.xs = o.x_stride; a.xoff = o.x_offset;
    a.kh = kh; a.kw = kw; a.pad = kh / 2;
    a.cout = o.cout; a.ys = o.y_stride; a.yoff = o.y_offset; a.relu = relu ? 1 : 0;
    a.m = m;
    dim3 grid((unsigned)bm, cout_pad / bn);
    st_timed t(ctx, ST_K_CONV);
    if (f32) {
      if (bn == 128) hipLaunchKernelGGL((k_conv_nhwc_f32<128, 16, 1, 1>), grid, dim3(256), 0, ctx->stream, a);
      else hipLaunchKernelGGL((k_conv_nhwc_f32<64, 16, 1, 1>), grid, dim3(256), 0, ctx->stream, a);
    } else {
      if (bn == 128) hipLaunchKernelGGL((k_conv_nhwc_bf16x3<128>), grid, dim3(256), 0, ctx->stream, a, (const unsigned*)o.w);
      else hipLaunchKernelGGL((k_conv_nhwc_bf16x3<64>), grid, dim3(256), 0, ctx->stream, a, (const unsigned*)o.w);
    }
    ST_HIP(ctx, hipGetLastError());
  }
  return ST_OK;
}
}  // namespace

ST_EXPORT int st_conv2d_nhwc_f32_tiled(st_ctx* ctx, const float* x_dev, int n, int h, int w, int cin, int x_stride, int x_offset,
                                       const float* w_dev, const void* wt_dev, const float* bias_dev, int kh, int kw, int cout, int cout_pad,
                                       int relu, float* y_dev, int y_stride, int y_offset) {
  ST_TRY(st_enter(ctx));
  const st_conv_operands o{x_dev, x_stride, x_offset, w_dev, wt_dev, bias_dev, cout, y_dev, y_stride, y_offset};
  return conv_launch(ctx, true, n, h, w, cin, kh, kw, cout_pad, relu, &o, 1);
}

ST_EXPORT int st_conv2d_nhwc_f32(st_ctx* ctx, const float* x_dev, int n, int h, int w, int cin, int x_stride, int x_offset,
                                 const float* w_dev, const float* bias_dev, int kh, int kw, int cout, int cout_pad, int relu,
                                 float* y_dev, int y_stride, int y_offset) {
  return st_conv2d_nhwc_f32_tiled(ctx, x_dev, n, h, w, cin, x_stride, x_offset, w_dev, nullptr, bias_dev, kh, kw, cout, cout_pad, relu, y_dev,
                                  y_stride, y_offset);
}

ST_EXPORT int st_conv2d_nhwc_f32_pair(st_ctx* ctx, int n, int h, int w, int cin, int kh, int kw, int cout_pad, int relu,
                                      const st_conv_operands* a, const st_conv_operands* b) {
  ST_TRY(st_enter(ctx));
  if (!a || !b) return st_set_error(ctx, ST_ERR_INVALID, "conv2d: null operands");
  const st_conv_operands ops[2] = {*a, *b};
  return conv_launch(ctx, true, n, h, w, cin, kh, kw, cout_pad, relu, ops, 2);
}

ST_EXPORT int st_conv2d_nhwc_bf16x3_pair(st_ctx* ctx, int n, int h, int w, int cin, int kh, int kw, int cout_pad, int relu,
                                         const st_conv_operands* a, const st_conv_operands* b) {
  ST_TRY(st_enter(ctx));
  if (!a || !b) return st_set_error(ctx, ST_ERR_INVALID, "conv2d: null operands");
  const st_conv_operands ops[2] = {*a, *b};
  return conv_launch(ctx, false, n, h, w, cin, kh, kw, cout_pad, relu, ops, 2);
}

ST_EXPORT long long st_conv_f32_tile_bytes(int cout_pad, int kh, int kw, int cin) {
  if (cout_pad <= 0 || kh <= 0 || kw <= 0 || cin <= 0 || !conv_tile_weights(kh, kw, cout_pad, cin)) return 0;
  return (long long)cout_pad * kh * kw * cin * 4;
}

ST_EXPORT int st_conv_pack_weights_f32_tile(st_ctx* ctx, const float* w_dev, int cout_pad, int kh, int kw, int cin, void* out_dev) {
  ST_TRY(st_enter(ctx));
  if (!w_dev || !out_dev || cout_pad <= 0 || kh <= 0 || kw <= 0 || cin <= 0 || ((uintptr_t)out_dev & 15))
    return st_set_error(ctx, ST_ERR_INVALID, "conv pack: bad arguments (16-byte aligned output)");
  if (!conv_tile_weights(kh, kw, cout_pad, cin))
    return st_set_error(ctx, ST_ERR_UNSUPPORTED, "conv pack: the tile order exists for 3x3 / 7x7 layers with cout_pad a multiple of 128 and cin a multiple of 16 (st_conv_f32_tile_bytes returns 0 otherwise)");
  const long long total = (long long)cout_pad * kh * kw * cin;
  long long bx = (total + 255) / 256;
  if (bx > 65536) bx = 65536;
  st_conv_tile_pack_f32(ctx, (unsigned)bx, w_dev, cout_pad, kh * kw, cin, (float*)out_dev);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

ST_EXPORT long long st_conv_bf16x3_packed_bytes(int cout_pad, int kh, int kw, int cin) {
  if (cout_pad <= 0 || kh <= 0 || kw <= 0 || cin <= 0) return 0;
  const long long one = (long long)cout_pad * kh * kw * cin * 6;
  return conv_tile_weights(kh, kw, cout_pad, cin) ? 2 * one : one;   // + the tile-order copy for k_conv_tile_bf16x3
}

ST_EXPORT int st_conv_pack_weights_bf16x3_n(st_ctx* ctx, const float* w_dev, int cout_pad, int kh, int kw, int cin, void* out_dev,
                                            size_t out_bytes) {
  ST_TRY(st_enter(ctx));
  if (!w_dev || !out_dev || cout_pad <= 0 || kh <= 0 || kw <= 0 || cin <= 0 || cin % 16 || ((uintptr_t)out_dev & 15))
    return st_set_error(ctx, ST_ERR_INVALID, "conv pack: bad arguments (cin a multiple of 16, 16-byte aligned output)");
  if ((long long)out_bytes < st_conv_bf16x3_packed_bytes(cout_pad, kh, kw, cin))
    return st_set_error(ctx, ST_ERR_INVALID, "conv pack: the output buffer is smaller than st_conv_bf16x3_packed_bytes() for this layer");
  const long long total = (long long)cout_pad * kh * kw * (cin / 2);
  long long bx = (total + 255) / 256;
  if (bx > 65536) bx = 65536;
  hipLaunchKernelGGL(k_pack_weights_bf16x3, dim3((unsigned)bx), dim3(256), 0, ctx->stream, w_dev, cout_pad, kh * kw, cin, (unsigned*)out_dev);
  if (conv_tile_weights(kh, kw, cout_pad, cin))
    st_conv_tile_pack_bf16x3(ctx, (unsigned)bx, w_dev, cout_pad, kh * kw, cin, (unsigned*)out_dev + (size_t)total * 3);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

// The entry point without a size: the caller vouches for st_conv_bf16x3_packed_bytes() bytes behind out_dev.
ST_EXPORT int st_conv_pack_weights_bf16x3(st_ctx* ctx, const float* w_dev, int cout_pad, int kh, int kw, int cin, void* out_dev) {
  const long long need = st_conv_bf16x3_packed_bytes(cout_pad, kh, kw, cin);
  return st_conv_pack_weights_bf16x3_n(ctx, w_dev, cout_pad, kh, kw, cin, out_dev, need > 0 ? (size_t)need : 0);
}

ST_EXPORT int st_conv2d_nhwc_bf16x3(st_ctx* ctx, const float* x_dev, int n, int h, int w, int cin, int x_stride, int x_offset,
                                    const void* w3_dev, const float* bias_dev, int kh, int kw, int cout, int cout_pad, int relu,
                                    float* y_dev, int y_stride, int y_offset) {
  ST_TRY(st_enter(ctx));
  const st_conv_operands o{x_dev, x_stride, x_offset, w3_dev, nullptr, bias_dev, cout, y_dev, y_stride, y_offset};
  return conv_launch(ctx, false, n, h, w, cin, kh, kw, cout_pad, relu, &o, 1);
}

ST_EXPORT int st_maxpool2_nhwc_f32(st_ctx* ctx, const float* x_dev, int n, int h, int w, int c, int x_stride, float* y_dev,
                                   int y_stride) {
  ST_TRY(st_enter(ctx));
  if (!x_dev || !y_dev || n <= 0 || h < 2 || w < 2 || c <= 0 || c % 4 || x_stride % 4 || y_stride % 4 || c > x_stride || c > y_stride ||
      ((uintptr_t)x_dev & 15) || ((uintptr_t)y_dev & 15))
    return st_set_error(ctx, ST_ERR_INVALID, "maxpool2: bad arguments (channel counts and strides in multiples of 4, 16-byte aligned buffers)");
  PoolArgs a;
  a.x = x_dev; a.y = y_dev; a.n = n; a.h = h; a.wd = w; a.c = c; a.xs = x_stride; a.ys = y_stride;
  const long long total = (long long)n * (h / 2) * (w / 2) * (c / 4);
  long long bx = (total + 255) / 256;
  if (bx > 65536) bx = 65536;
  st_timed t(ctx, ST_K_CONV);
  hipLaunchKernelGGL(k_maxpool2_nhwc_f32, dim3((unsigned)bx), dim3(256), 0, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

ST_EXPORT int st_planar_to_nhwc_f32(st_ctx* ctx, const float* x_dev, int n, int c, int h, int w, float* y_dev, int y_stride) {
  ST_TRY(st_enter(ctx));
  if (!x_dev || !y_dev || n <= 0 || c <= 0 || h <= 0 || w <= 0 || y_stride < c || y_stride % 4 || ((uintptr_t)y_dev & 15))
    return st_set_error(ctx, ST_ERR_INVALID, "planar_to_nhwc: bad arguments (output channel count a multiple of 4 >= c, 16-byte aligned)");
  PlanarArgs a;
  a.x = x_dev; a.y = y_dev; a.n = n; a.c = c; a.h = h; a.wd = w; a.cs = y_stride;
  const long long total = (long long)n * h * w * (y_stride / 4);
  long long bx = (total + 255) / 256;
  if (bx > 65536) bx = 65536;
  st_timed t(ctx, ST_K_CONV);
  hipLaunchKernelGGL(k_planar_to_nhwc, dim3((unsigned)bx), dim3(256), 0, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}
