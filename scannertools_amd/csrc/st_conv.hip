// Convolution stack of the pose network (BASELINE config 5) for gfx950:
// stride-1 "same" convolutions (3x3, 7x7, 1x1) + bias + ReLU as implicit GEMMs on the matrix cores,
// and the 2x2 max pooling between the VGG blocks.  This is the only dense contraction in the
// repository's scope, hence the only MFMA code.
//
// What it replaces: the Caffe forward pass behind the reference's CPM2 op
// (/root/reference/scannertools_caffe/scannertools_caffe_cpp/cpm2_kernel.cpp:8-52 -> CaffeKernel::execute,
// caffe_kernel.cpp) for the layers of the OpenPose COCO model.  The reference
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

#include "st_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// two floats -> packed bf16 pairs of the three terms
__device__ __forceinline__ void split3(float v0, float v1, unsigned& hi, unsigned& mid, unsigned& lo) {
  f32x2 v = {v0, v1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  v.x = v0 - __uint_as_float(hi << 16);
  v.y = v1 - __uint_as_float(hi & 0xffff0000u);
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  v.x -= __uint_as_float(mid << 16);
  v.y -= __uint_as_float(mid & 0xffff0000u);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

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

// ---------------------------------------------------------------------------------------------------------------
// bf16x3 on a SPATIAL tile (round 4).  k_conv_nhwc_bf16x3 pulls, per 16-channel slice of one tap, 8 KB of float32
// pixels and 12 KB of split weights through the L1 for a 128 x 128 tile -- 31 B per clock and CU at full matrix rate,
// more than a CU's miss queue sustains; its matrix pipe is busy 43 % of the time.  A 7x7 layer fetches
// every input pixel 49 times that way.  Here a workgroup owns TH x TW <= 32 NW output pixels of ONE image and 128 output
// channels:
//   * per 16-channel slice the input REGION (TH + K - 1) x (TW + K - 1) of the tile is split into its three bf16 planes
//     ONCE and kept in LDS (NW = 4: <= 400 pixels x 96 B, two buffers = 75 KB, two workgroups of 256 threads per CU); the K x K
//     taps of the slice read their A operands from it at a tap-dependent offset -- no global traffic, no split and no
//     barrier between taps;
//   * the B operands (weights) never touch LDS: st_conv_pack_weights_bf16x3 also writes them in the instruction's own
//     operand order ([cout block][slice][tap][32-column tile][split][lane] x 16 bytes), so a wave fetches the 6 KB of a tap
//     with six fully coalesced 16-byte loads straight into registers, TWO taps ahead (a tap's 768 matrix cycles do not cover
//     an L2 round trip); the waves that share them hit L1;
//   * the next slice's region arrives one (pixel, channel quad) item per thread at a time and is split / stored into the other
//     LDS buffer between the taps' matrix instructions; ONE barrier per slice (49 or 9 taps x 24 MFMAs per wave);
//   * inside a tap every matrix instruction is followed by one pinned piece of the side work (see the main loop).
// L1 fills per executed flop fall 1.7-fold with 128 x 128 tiles (3.3-fold with 256 x 128: 12 KB of weights per tap and slice
// either way, the region's 0.4-2 KB amortised over the taps).  Accumulation order of an output: slices outer, taps inner, the six terms as in
// k_conv_nhwc_bf16x3 -- independent of the tile shape, so the tiling never changes a bit, and the order the per-tap kernel
// walks K in too: the two kernels give the same bits and the launcher may pick either.
// ---------------------------------------------------------------------------------------------------------------
constexpr int CT_RPMAX = 800;   // region pixels per LDS buffer
constexpr int CT_ITEMS = (CT_RPMAX * 4 + 511) / 512;   // (pixel, channel quad) items per thread and slice

// One launch may carry TWO convolutions of the same geometry (grid.z: the two branches of a stage of the pose network, which
// read different activations with different weights): at the reference's five frames per call a 7x7 layer is 160 workgroups
// for 256 CUs, the pair 320.
struct ConvTileOperands {
  const float* x;
  const unsigned* w3t;   // tile-order weights
  const float* bias;
  float* y;
  int xs, xoff, cout, ys, yoff;
};
struct ConvTileArgs {
  ConvTileOperands op[2];
  int n, h, wd, cin, pad, relu;
  int th, tw, rw, rp;        // tile rows / columns, region columns, region pixels
  int tiles_x, tiles_y;      // tiles per image
};

__global__ __launch_bounds__(256) void k_pack_weights_bf16x3_tile(const float* __restrict__ w, int cout_pad, int taps, int cin,
                                                                  unsigned* __restrict__ out) {
  // thread = (output channel, tap, channel pair)
  const long long total = (long long)cout_pad * taps * (cin / 2);
  const int S = cin / 16;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cp = (int)(i % (cin / 2));
    const long long row = i / (cin / 2);
    const int tap = (int)(row % taps), co = (int)(row / taps);
    const float* src = w + row * cin + 2 * cp;
    unsigned h, m, l;
    split3(src[0], src[1], h, m, l);
    const int c = 2 * cp, slice = c >> 4, k = c & 15;
    const int cb = co >> 7, j = (co >> 5) & 3, c31 = co & 31;
    const int lane = (k >> 3) * 32 + c31, d = (k & 7) >> 1;
    const size_t base = ((((size_t)cb * S + slice) * taps + tap) * 4 + j) * 3;
    out[((base + 0) * 64 + lane) * 4 + d] = h;
    out[((base + 1) * 64 + lane) * 4 + d] = m;
    out[((base + 2) * 64 + lane) * 4 + d] = l;
  }
}

// NW: waves per workgroup -- 4 (128 pixels, region <= 400 pixels, two workgroups per CU; the instance in use) or 8 (256 pixels,
// region <= 800, one per CU; level with it from 16 frames per call on, behind it below -- not instantiated); the same bits, the
// accumulation order does not depend on the tile
// MT: 32-pixel instruction tiles per wave -- 2 (a wave owns 64 pixels x 64 channels) or 1 (32 x 64: half the serial work per
// wave and twice the workgroups, for launches of a few frames)
template <int KS, int NW, int MT = 2>
__global__ __launch_bounds__(NW * 64, 2) void k_conv_tile_bf16x3(ConvTileArgs a) {
  const ConvTileOperands& o = a.op[blockIdx.z];
  constexpr int T = KS * KS, THREADS = NW * 64;
  __shared__ unsigned Ar[2][3][(NW == 8 ? CT_RPMAX : CT_RPMAX / 2) * 8];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, l31 = lane & 31, lk = lane >> 5;
  int bt = blockIdx.x;
  const int txi = bt % a.tiles_x;
  bt /= a.tiles_x;
  const int tyi = bt % a.tiles_y, img = bt / a.tiles_y;
  const int y0 = tyi * a.th, x0 = txi * a.tw;
  const int S = a.cin / 16;
  const float* __restrict__ ximg = o.x + (size_t)img * a.h * a.wd * o.xs + o.xoff;

  // region pixel r, 16-byte half c -> dword offset inside a plane (32-byte rows, the halves swapped on rows 8..15 mod 16)
  auto swz = [](int r, int c) { return r * 8 + ((c ^ ((r >> 3) & 1)) << 2); };

  // this thread's items of a region: (pixel, channel quad), the same for every slice
  int goff[CT_ITEMS], loff[CT_ITEMS];
  unsigned exists = 0, inb = 0;
#pragma unroll
  for (int i = 0; i < CT_ITEMS; ++i) {
    const int e = i * THREADS + t, r = e >> 2, cq = e & 3;
    goff[i] = 0;
    loff[i] = 0;
    if (r < a.rp) {
      const int ry = r / a.rw, rx = r - ry * a.rw;
      const int yy = y0 - a.pad + ry, xx = x0 - a.pad + rx;
      exists |= 1u << i;
      loff[i] = swz(r, cq >> 1) + 2 * (cq & 1);
      if ((unsigned)yy < (unsigned)a.h && (unsigned)xx < (unsigned)a.wd) {
        inb |= 1u << i;
        goff[i] = (yy * a.wd + xx) * o.xs + 4 * cq;
      }
    }
  }
  // item i of the next slice's region is requested at tap ILOAD(i) and split / stored at tap ISTORE(i), between the taps' matrix
  // instructions (7x7: one float4 in flight per thread, five taps for the round trip; 3x3: two taps, three in flight)
  constexpr int IGAP = KS == 7 ? 7 : 1, ILAT = KS == 7 ? 5 : 2, NRG = KS == 7 ? 1 : 3;
  float4 rg[NRG];
  auto stash_from = [&](int buf, int i, float4 v) {
    if ((exists >> i) & 1) {
      unsigned h0, m0, l0, h1, m1, l1;
      split3(v.x, v.y, h0, m0, l0);
      split3(v.z, v.w, h1, m1, l1);
      *reinterpret_cast<uint2*>(&Ar[buf][0][0] + loff[i]) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(&Ar[buf][1][0] + loff[i]) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(&Ar[buf][2][0] + loff[i]) = make_uint2(l0, l1);
    }
  };

  // the wave's 32 MT pixels x 64 output channels: MT x 2 instruction tiles
  const int wm = (wv >> 1) * 32 * MT;
  const int npix = a.th * a.tw;
  int abase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int p = wm + 32 * i + l31;
    if (p >= npix) p = npix - 1;   // rows of the instruction tile beyond the spatial tile: computed, never stored
    const int ty = p / a.tw;
    abase[i] = ty * a.rw + (p - ty * a.tw);
  }
  // weights of (slice, tap) q for this wave: 6 x 1 KB, lane-contiguous
  const uint4* __restrict__ wq = reinterpret_cast<const uint4*>(o.w3t) + ((size_t)blockIdx.y * S * T) * 768 + (wv & 1) * 384 + lane;

  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 na[3][MT], nb[2][2][3];   // pixels of the next tap; weights of the next two taps (a tap's 768 matrix cycles do not cover an L2 round trip)
  auto loadB = [&](int stage, int q) {
    const uint4* __restrict__ src = wq + (size_t)q * 768;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int sp = 0; sp < 3; ++sp) nb[stage][j][sp] = src[(j * 3 + sp) * 64];
  };
  auto readA = [&](int buf, int tapoff) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      // recomputed per tap: hoisted out of the slice loop, the 2 T offsets do not fit the registers, and their scratch reloads
      // would wait (vmcnt) for the weight loads just issued
      int ab = abase[i];
      asm volatile("" : "+v"(ab));
      const int idx = ab + tapoff;
      const int o = swz(idx, lk);
#pragma unroll
      for (int sp = 0; sp < 3; ++sp) na[sp][i] = *reinterpret_cast<const uint4*>(&Ar[buf][sp][0] + o);
    }
  };

#pragma unroll
  for (int i = 0; i < CT_ITEMS; ++i)
    stash_from(0, i, (inb >> i) & 1 ? *reinterpret_cast<const float4*>(ximg + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f));
  const int nq = S * T;
  loadB(0, 0);
  if (nq > 1) loadB(1, 1);
  __syncthreads();
  readA(0, 0);
  for (int s = 0; s < S; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < S;   // uniform
#pragma clang loop unroll(full)
    for (int tap = 0; tap < T; ++tap) {
      bf16x8 af[3][MT], bfr[3][2];
#pragma unroll
      for (int sp = 0; sp < 3; ++sp) {
#pragma unroll
        for (int i = 0; i < MT; ++i) af[sp][i] = __builtin_bit_cast(bf16x8, na[sp][i]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bfr[sp][j] = __builtin_bit_cast(bf16x8, nb[0][j][sp]);
          nb[0][j][sp] = nb[1][j][sp];
        }
      }
      // The tap's 12 MT matrix instructions, each followed by ONE piece (two where there are more pieces than instructions) of
      // the work for later taps, pinned in this order (sched_barrier): a matrix instruction occupies the pipe for 32 cycles
      // after it issues, so the wave's own loads and address arithmetic placed behind it are free, while the same
      // instructions in one clump at the head of the tap leave the pipe to the SIMD's other wave alone.  Pieces: 0-5 the six
      // weight loads of tap + 2 (unconditional: the last two taps re-request the last tap's weights, so the code is
      // straight-line and the three operand sets rotate by renaming), then the 3 MT region reads of tap + 1, the request of a
      // region item of the next slice, its split and stores (5).
      const int q2 = s * T + tap + 2;
      const uint4* __restrict__ wsrc = wq + (size_t)(q2 < nq ? q2 : nq - 1) * 768;
      int ao[MT] = {};
      const bool item_load = tap % IGAP == 0 && tap / IGAP < CT_ITEMS;   // constants after unrolling
      const bool item_store = tap >= ILAT && (tap - ILAT) % IGAP == 0 && (tap - ILAT) / IGAP < CT_ITEMS;
      const int li = item_load ? tap / IGAP : 0, si = item_store ? (tap - ILAT) / IGAP : 0;
      unsigned h0 = 0, m0 = 0, l0 = 0, h1 = 0, m1 = 0, l1 = 0;
      constexpr int NMF = 12 * MT, PA = 6, PI = PA + 3 * MT, NPIECE = PI + 6;
      auto piece = [&](int p) {
        if (p < PA) {
          nb[1][p / 3][p % 3] = wsrc[p * 64];
        } else if (p < PI) {
          if (tap + 1 < T) {
            const int i2 = (p - PA) / 3, sp = (p - PA) % 3;
            if (sp == 0) {
              int rw = a.rw, ab = abase[i2];
              asm volatile("" : "+s"(rw));   // recomputed per tap: hoisted out of the slice loop the 2 T offsets do not fit the
              asm volatile("" : "+v"(ab));   // registers, and their scratch reloads would wait for the weight loads in flight
              ao[i2] = swz(ab + ((tap + 1) / KS) * rw + (tap + 1) % KS, lk);
            }
            na[sp][i2] = *reinterpret_cast<const uint4*>(&Ar[buf][sp][0] + ao[i2]);
          }
        } else if (p == PI) {
          if (item_load && more)
            rg[li % NRG] = (inb >> li) & 1 ? *reinterpret_cast<const float4*>(ximg + goff[li] + 16 * (s + 1)) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (item_store && more && ((exists >> si) & 1)) {
          const float4 v = rg[si % NRG];
          if (p == PI + 1) split3(v.x, v.y, h0, m0, l0);
          if (p == PI + 2) split3(v.z, v.w, h1, m1, l1);
          if (p == PI + 3) *reinterpret_cast<uint2*>(&Ar[buf ^ 1][0][0] + loff[si]) = make_uint2(h0, h1);
          if (p == PI + 4) *reinterpret_cast<uint2*>(&Ar[buf ^ 1][1][0] + loff[si]) = make_uint2(m0, m1);
          if (p == PI + 5) *reinterpret_cast<uint2*>(&Ar[buf ^ 1][2][0] + loff[si]) = make_uint2(l0, l1);
        }
      };
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NMF; ++k) {
        const int term = k / (2 * MT), i = (k >> 1) % MT, j = k & 1;
        const int sa = (term == 0 || term == 1 || term == 4) ? 0 : (term == 5 ? 2 : 1);
        const int sb = (term == 0 || term == 2 || term == 5) ? 0 : (term == 4 ? 2 : 1);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[sa][i], bfr[sb][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // pieces in order, spread evenly over the instructions (dependent ones -- split before store -- stay in sequence)
#pragma unroll
        for (int p2 = k * NPIECE / NMF; p2 < (k + 1) * NPIECE / NMF; ++p2) piece(p2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    if (more) readA(buf ^ 1, 0);
  }

  // bias (+ ReLU), float32 store of the pixels inside the tile and the image
  const int wn = (wv & 1) * 64;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (p >= npix) continue;
      const int ty = p / a.tw, tx = p - ty * a.tw;
      const int yy = y0 + ty, xx = x0 + tx;
      if (yy >= a.h || xx >= a.wd) continue;
      float* __restrict__ yp = o.y + ((size_t)((size_t)img * a.h + yy) * a.wd + xx) * o.ys + o.yoff;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nn = blockIdx.y * 128 + wn + 32 * j + l31;
        if (nn < o.cout) {
          float v = acc[i][j][r] + o.bias[nn];
          if (a.relu) v = v > 0.f ? v : 0.f;
          yp[nn] = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The spatial tile for the float32 instruction (v_mfma_f32_32x32x2_f32): same workgroup shape, region, weight streaming and
// pinned side work as k_conv_tile_bf16x3; what differs is the operand layout.  LDS: the region channel-major,
// [buffer][16 channels][818 (418 for the 4-wave instance)] floats (a lane's A operand of step kk is ONE float, channel 2 kk + (lane >> 5) of its pixel:
// 32 consecutive lanes read 32 consecutive floats; the plane stride 818 = 2 mod 16 spreads an item's four channel stores of
// eight pixels over 32 banks).  Weights in operand order: [cout block][slice][tap][32-column tile][4-step group][lane] x 4 floats
// (st_conv_pack_weights_f32_tile), four 16-byte loads per wave and tap, one tap ahead (a tap is 32 instructions of 64 cycles).
// Accumulation order of an output: slices outer, taps inner, channels ascending inside a slice -- a k-ordered fmaf chain as
// in k_conv_nhwc_f32 (the same walk there: same bits), independent of the tile shape.
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_pack_weights_f32_tile(const float* __restrict__ w, int cout_pad, int taps, int cin,
                                                               float* __restrict__ out) {
  const long long total = (long long)cout_pad * taps * cin;
  const int S = cin / 16;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cin);
    const long long row = i / cin;
    const int tap = (int)(row % taps), co = (int)(row / taps);
    const int slice = c >> 4, k = c & 15, kk = k >> 1, lk = k & 1;
    const int cb = co >> 7, j = (co >> 5) & 3, c31 = co & 31;
    const size_t base = (((((size_t)cb * S + slice) * taps + tap) * 4 + j) * 2 + (kk >> 2)) * 64 + (lk * 32 + c31);
    out[base * 4 + (kk & 3)] = w[i];
  }
}

template <int KS, int NW, int MT = 2>
__global__ __launch_bounds__(NW * 64, 2) void k_conv_tile_f32(ConvTileArgs a) {
  const ConvTileOperands& o = a.op[blockIdx.z];
  constexpr int T = KS * KS, THREADS = NW * 64, CTF_RPS = NW == 8 ? 818 : 418;   // plane strides = 2 mod 16
  __shared__ float Af[2][16][CTF_RPS];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, l31 = lane & 31, lk = lane >> 5;
  int bt = blockIdx.x;
  const int txi = bt % a.tiles_x;
  bt /= a.tiles_x;
  const int tyi = bt % a.tiles_y, img = bt / a.tiles_y;
  const int y0 = tyi * a.th, x0 = txi * a.tw;
  const int S = a.cin / 16;
  const float* __restrict__ ximg = o.x + (size_t)img * a.h * a.wd * o.xs + o.xoff;

  int goff[CT_ITEMS], loff[CT_ITEMS];
  unsigned exists = 0, inb = 0;
#pragma unroll
  for (int i = 0; i < CT_ITEMS; ++i) {
    const int e = i * THREADS + t, r = e >> 2, cq = e & 3;
    goff[i] = 0;
    loff[i] = 0;
    if (r < a.rp) {
      const int ry = r / a.rw, rx = r - ry * a.rw;
      const int yy = y0 - a.pad + ry, xx = x0 - a.pad + rx;
      exists |= 1u << i;
      loff[i] = 4 * cq * CTF_RPS + r;
      if ((unsigned)yy < (unsigned)a.h && (unsigned)xx < (unsigned)a.wd) {
        inb |= 1u << i;
        goff[i] = (yy * a.wd + xx) * o.xs + 4 * cq;
      }
    }
  }
  constexpr int IGAP = KS == 7 ? 7 : 1, ILAT = KS == 7 ? 5 : 2, NRG = KS == 7 ? 1 : 3;
  float4 rg[NRG];
  auto stash_from = [&](int buf, int i, float4 v) {
    if ((exists >> i) & 1) {
      float* __restrict__ d = &Af[buf][0][0] + loff[i];
      d[0] = v.x; d[CTF_RPS] = v.y; d[2 * CTF_RPS] = v.z; d[3 * CTF_RPS] = v.w;
    }
  };

  const int wm = (wv >> 1) * 32 * MT;
  const int npix = a.th * a.tw;
  int abase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int p = wm + 32 * i + l31;
    if (p >= npix) p = npix - 1;
    const int ty = p / a.tw;
    abase[i] = ty * a.rw + (p - ty * a.tw) + lk * CTF_RPS;
  }
  const uint4* __restrict__ wq = reinterpret_cast<const uint4*>(o.w3t) + ((size_t)blockIdx.y * S * T) * 512 + (wv & 1) * 256 + lane;

  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float na[MT][8];     // pixels of the next tap: [instruction tile][step]
  uint4 nb[2][2];      // weights of the next tap: [column tile][4-step group]
  auto loadB = [&](int q) {
    const uint4* __restrict__ src = wq + (size_t)q * 512;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 2; ++g) nb[j][g] = src[(j * 2 + g) * 64];
  };
  auto readA = [&](int buf, int tapoff) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int ab = abase[i];
      asm volatile("" : "+v"(ab));
      const float* __restrict__ src = &Af[buf][0][0] + ab + tapoff;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) na[i][kk] = src[2 * kk * CTF_RPS];
    }
  };

#pragma unroll
  for (int i = 0; i < CT_ITEMS; ++i)
    stash_from(0, i, (inb >> i) & 1 ? *reinterpret_cast<const float4*>(ximg + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f));
  const int nq = S * T;
  loadB(0);
  __syncthreads();
  readA(0, 0);
  for (int s = 0; s < S; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < S;   // uniform
#pragma clang loop unroll(full)
    for (int tap = 0; tap < T; ++tap) {
      float af[MT][8], bfr[2][8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i][kk] = na[i][kk];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint4 v = nb[j][kk >> 2];
          bfr[j][kk] = __uint_as_float((kk & 3) == 0 ? v.x : (kk & 3) == 1 ? v.y : (kk & 3) == 2 ? v.z : v.w);
        }
      }
      // 16 MT matrix instructions with the pieces of the work for the next tap spread between them in order
      // (k_conv_tile_bf16x3): the four weight loads of tap + 1 (unconditional, clamped), the 8 MT region reads of tap + 1, the
      // request of a region item of the next slice, its four stores
      const int q1 = s * T + tap + 1;
      const uint4* __restrict__ wsrc = wq + (size_t)(q1 < nq ? q1 : nq - 1) * 512;
      const float* asrc[MT] = {};
      const bool item_load = tap % IGAP == 0 && tap / IGAP < CT_ITEMS;   // constants after unrolling
      const bool item_store = tap >= ILAT && (tap - ILAT) % IGAP == 0 && (tap - ILAT) / IGAP < CT_ITEMS;
      const int li = item_load ? tap / IGAP : 0, si = item_store ? (tap - ILAT) / IGAP : 0;
      constexpr int NMF = 16 * MT, PA = 4, PI = PA + 8 * MT, NPIECE = PI + 2;
      auto piece = [&](int p) {
        if (p < PA) {
          nb[p >> 1][p & 1] = wsrc[p * 64];
        } else if (p < PI) {
          if (tap + 1 < T) {
            const int i2 = (p - PA) >> 3, k2 = (p - PA) & 7;
            if (k2 == 0) {
              int rw = a.rw, ab = abase[i2];
              asm volatile("" : "+s"(rw));   // recomputed per tap (see k_conv_tile_bf16x3)
              asm volatile("" : "+v"(ab));
              asrc[i2] = &Af[buf][0][0] + ab + ((tap + 1) / KS) * rw + (tap + 1) % KS;
            }
            na[i2][k2] = asrc[i2][2 * k2 * CTF_RPS];
          }
        } else if (p == PI) {
          if (item_load && more)
            rg[li % NRG] = (inb >> li) & 1 ? *reinterpret_cast<const float4*>(ximg + goff[li] + 16 * (s + 1)) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          if (item_store && more) stash_from(buf ^ 1, si, rg[si % NRG]);
        }
      };
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NMF; ++k) {
        const int kk = k / (2 * MT), i = (k >> 1) % MT, j = k & 1;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bfr[j][kk], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p2 = k * NPIECE / NMF; p2 < (k + 1) * NPIECE / NMF; ++p2) piece(p2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    if (more) readA(buf ^ 1, 0);
  }

  const int wn = (wv & 1) * 64;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (p >= npix) continue;
      const int ty = p / a.tw, tx = p - ty * a.tw;
      const int yy = y0 + ty, xx = x0 + tx;
      if (yy >= a.h || xx >= a.wd) continue;
      float* __restrict__ yp = o.y + ((size_t)((size_t)img * a.h + yy) * a.wd + xx) * o.ys + o.yoff;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nn = blockIdx.y * 128 + wn + 32 * j + l31;
        if (nn < o.cout) {
          float v = acc[i][j][r] + o.bias[nn];
          if (a.relu) v = v > 0.f ? v : 0.f;
          yp[nn] = v;
        }
      }
    }
}

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
    if (f32) {
      if (kh == 7 && nw == 8) hipLaunchKernelGGL((k_conv_tile_f32<7, 8>), tgrid, dim3(512), 0, ctx->stream, ta);
      else if (kh == 7 && nw == 41) hipLaunchKernelGGL((k_conv_tile_f32<7, 4, 1>), tgrid, dim3(256), 0, ctx->stream, ta);
      else if (kh == 7) hipLaunchKernelGGL((k_conv_tile_f32<7, 4>), tgrid, dim3(256), 0, ctx->stream, ta);
      else if (nw == 8) hipLaunchKernelGGL((k_conv_tile_f32<3, 8>), tgrid, dim3(512), 0, ctx->stream, ta);
      else if (nw == 41) hipLaunchKernelGGL((k_conv_tile_f32<3, 4, 1>), tgrid, dim3(256), 0, ctx->stream, ta);
      else hipLaunchKernelGGL((k_conv_tile_f32<3, 4>), tgrid, dim3(256), 0, ctx->stream, ta);
    } else {
      if (kh == 7 && nw == 41) hipLaunchKernelGGL((k_conv_tile_bf16x3<7, 4, 1>), tgrid, dim3(256), 0, ctx->stream, ta);
      else if (kh == 7) hipLaunchKernelGGL((k_conv_tile_bf16x3<7, 4>), tgrid, dim3(256), 0, ctx->stream, ta);
      else if (nw == 41) hipLaunchKernelGGL((k_conv_tile_bf16x3<3, 4, 1>), tgrid, dim3(256), 0, ctx->stream, ta);
      else hipLaunchKernelGGL((k_conv_tile_bf16x3<3, 4>), tgrid, dim3(256), 0, ctx->stream, ta);
    }
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
  hipLaunchKernelGGL(k_pack_weights_f32_tile, dim3((unsigned)bx), dim3(256), 0, ctx->stream, w_dev, cout_pad, kh * kw, cin, (float*)out_dev);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

ST_EXPORT long long st_conv_bf16x3_packed_bytes(int cout_pad, int kh, int kw, int cin) {
  if (cout_pad <= 0 || kh <= 0 || kw <= 0 || cin <= 0) return 0;
  const long long one = (long long)cout_pad * kh * kw * cin * 6;
  return conv_tile_weights(kh, kw, cout_pad, cin) ? 2 * one : one;   // + the tile-order copy for k_conv_tile_bf16x3
}

ST_EXPORT int st_conv_pack_weights_bf16x3(st_ctx* ctx, const float* w_dev, int cout_pad, int kh, int kw, int cin, void* out_dev) {
  ST_TRY(st_enter(ctx));
  if (!w_dev || !out_dev || cout_pad <= 0 || kh <= 0 || kw <= 0 || cin <= 0 || cin % 16 || ((uintptr_t)out_dev & 15))
    return st_set_error(ctx, ST_ERR_INVALID, "conv pack: bad arguments (cin a multiple of 16, 16-byte aligned output)");
  const long long total = (long long)cout_pad * kh * kw * (cin / 2);
  long long bx = (total + 255) / 256;
  if (bx > 65536) bx = 65536;
  hipLaunchKernelGGL(k_pack_weights_bf16x3, dim3((unsigned)bx), dim3(256), 0, ctx->stream, w_dev, cout_pad, kh * kw, cin, (unsigned*)out_dev);
  if (conv_tile_weights(kh, kw, cout_pad, cin))
    hipLaunchKernelGGL(k_pack_weights_bf16x3_tile, dim3((unsigned)bx), dim3(256), 0, ctx->stream, w_dev, cout_pad, kh * kw, cin,
                       (unsigned*)out_dev + (size_t)total * 3);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
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
