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
// The next slice's global loads are in flight while the current one is multiplied (registers -> other
// LDS buffer after the MFMAs; one barrier per slice).
#include <cstring>

#include "st_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CV_BM = 128, CV_BK = 16, CV_PAD = 4;

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

template <int BN>
__global__ __launch_bounds__(256) void k_conv_nhwc_f32(ConvArgs a) {
  constexpr int NB4 = BN * CV_BK / 4 / 256;  // float4 loads of the weight tile per thread (2 or 1)
  __shared__ float As[2][CV_BK][CV_BM + CV_PAD];
  __shared__ float Bs[2][CV_BK][BN + CV_PAD];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const long long m0 = (long long)blockIdx.x * CV_BM;
  const int n0 = blockIdx.y * BN;

  // this thread's pixel of the A tile (fixed over the K walk) and its channel quads
  const int pm = t & (CV_BM - 1), cqa = t >> 7;  // quads cqa, cqa + 2
  const long long gm = m0 + pm;
  const bool mvalid = gm < a.m;
  int px = 0, py = 0, pn = 0;
  if (mvalid) {
    px = (int)(gm % a.wd);
    const long long r = gm / a.wd;
    py = (int)(r % a.h);
    pn = (int)(r / a.h);
  }
  // weight tile: output channel and channel quads
  const int nb = t & (BN - 1), cqb = t / BN;  // BN = 128: quads cqb, cqb + 2; BN = 64: quad cqb (0..3)
  const size_t wrow = (size_t)a.kh * a.kw * a.cin;
  const float* __restrict__ wbase = a.w + (size_t)(n0 + nb) * wrow;

  const int cslices = a.cin / CV_BK;
  const int nslices = a.kh * a.kw * cslices;

  float4 ra[2], rb[2];
  auto fetch = [&](int s) {
    const int kpos = s / cslices, c0 = (s - kpos * cslices) * CV_BK;
    const int ky = kpos / a.kw, kx = kpos - ky * a.kw;
    const int yy = py + ky - a.pad, xx = px + kx - a.pad;
    const bool inb = mvalid && (unsigned)yy < (unsigned)a.h && (unsigned)xx < (unsigned)a.wd;
    const float* __restrict__ src = a.x + ((size_t)((size_t)pn * a.h + (inb ? yy : 0)) * a.wd + (inb ? xx : 0)) * a.xs + a.xoff + c0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      ra[j] = inb ? *reinterpret_cast<const float4*>(src + 4 * (cqa + 2 * j)) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* __restrict__ wsrc = wbase + (size_t)kpos * a.cin + c0;
#pragma unroll
    for (int j = 0; j < NB4; ++j) rb[j] = *reinterpret_cast<const float4*>(wsrc + 4 * (cqb + (BN == 128 ? 2 : 0) * j));
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = 4 * (cqa + 2 * j);
      As[buf][k][pm] = ra[j].x; As[buf][k + 1][pm] = ra[j].y; As[buf][k + 2][pm] = ra[j].z; As[buf][k + 3][pm] = ra[j].w;
    }
#pragma unroll
    for (int j = 0; j < NB4; ++j) {
      const int k = 4 * (cqb + (BN == 128 ? 2 : 0) * j);
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

  fetch(0);
  stash(0);
  __syncthreads();
  const int l31 = lane & 31, lk = lane >> 5;
  for (int s = 0; s < nslices; ++s) {
    const int buf = s & 1;
    if (s + 1 < nslices) fetch(s + 1);
#pragma unroll
    for (int kk = 0; kk < CV_BK; kk += 2) {
      float af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = As[buf][kk + lk][wm + 32 * i + l31];
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = Bs[buf][kk + lk][wn + 32 * j + l31];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < nslices) stash(buf ^ 1);
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

// planar (CPM2Input's output) -> NHWC with zero pad channels (the first layer's operand)
__global__ __launch_bounds__(256) void k_planar_to_nhwc(PlanarArgs a) {
  const long long total = (long long)a.n * a.h * a.wd;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long hw = (long long)a.h * a.wd;
    const int n = (int)(i / hw);
    const long long p = i - (long long)n * hw;
    float* __restrict__ d = a.y + (size_t)i * a.cs;
    for (int c = 0; c < a.cs; ++c) d[c] = c < a.c ? a.x[((size_t)n * a.c + c) * hw + p] : 0.f;
  }
}

}  // namespace

ST_EXPORT int st_conv2d_nhwc_f32(st_ctx* ctx, const float* x_dev, int n, int h, int w, int cin, int x_stride, int x_offset,
                                 const float* w_dev, const float* bias_dev, int kh, int kw, int cout, int cout_pad, int relu,
                                 float* y_dev, int y_stride, int y_offset) {
  ST_TRY(st_enter(ctx));
  if (!x_dev || !w_dev || !bias_dev || !y_dev || n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0)
    return st_set_error(ctx, ST_ERR_INVALID, "conv2d: bad arguments");
  if (cin % CV_BK || x_offset % 4 || x_stride % 4 || x_offset + cin > x_stride || ((uintptr_t)x_dev & 15) || ((uintptr_t)w_dev & 15))
    return st_set_error(ctx, ST_ERR_INVALID, "conv2d: input channels must be a multiple of 16 inside a 16-byte aligned buffer (cin=%d stride=%d offset=%d)", cin, x_stride, x_offset);
  if (kh != kw || !(kh & 1) || kh > 7) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "conv2d: %dx%d kernels (odd square kernels up to 7 are implemented)", kh, kw);
  const int bn = cout_pad % 128 == 0 ? 128 : 64;
  if (cout_pad < cout || cout_pad % 64) return st_set_error(ctx, ST_ERR_INVALID, "conv2d: cout_pad must be a multiple of 64 >= cout");
  if (y_offset + cout > y_stride) return st_set_error(ctx, ST_ERR_INVALID, "conv2d: output slice exceeds the buffer's channel count");
  ConvArgs a;
  a.x = x_dev; a.w = w_dev; a.bias = bias_dev; a.y = y_dev;
  a.n = n; a.h = h; a.wd = w; a.cin = cin; a.xs = x_stride; a.xoff = x_offset;
  a.kh = kh; a.kw = kw; a.pad = kh / 2;
  a.cout = cout; a.ys = y_stride; a.yoff = y_offset; a.relu = relu ? 1 : 0;
  a.m = (long long)n * h * w;
  const long long bm = (a.m + CV_BM - 1) / CV_BM;
  if (bm > 2147483647LL) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "conv2d: too many output pixels");
  dim3 grid((unsigned)bm, cout_pad / bn);
  st_timed t(ctx, ST_K_CONV);
  if (bn == 128) hipLaunchKernelGGL(k_conv_nhwc_f32<128>, grid, dim3(256), 0, ctx->stream, a);
  else hipLaunchKernelGGL(k_conv_nhwc_f32<64>, grid, dim3(256), 0, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

ST_EXPORT int st_maxpool2_nhwc_f32(st_ctx* ctx, const float* x_dev, int n, int h, int w, int c, int x_stride, float* y_dev,
                                   int y_stride) {
  ST_TRY(st_enter(ctx));
  if (!x_dev || !y_dev || n <= 0 || h < 2 || w < 2 || c <= 0 || c % 4 || x_stride % 4 || y_stride % 4 || c > x_stride || c > y_stride)
    return st_set_error(ctx, ST_ERR_INVALID, "maxpool2: bad arguments");
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
  if (!x_dev || !y_dev || n <= 0 || c <= 0 || h <= 0 || w <= 0 || y_stride < c) return st_set_error(ctx, ST_ERR_INVALID, "planar_to_nhwc: bad arguments");
  PlanarArgs a;
  a.x = x_dev; a.y = y_dev; a.n = n; a.c = c; a.h = h; a.wd = w; a.cs = y_stride;
  const long long total = (long long)n * h * w;
  long long bx = (total + 255) / 256;
  if (bx > 65536) bx = 65536;
  hipLaunchKernelGGL(k_planar_to_nhwc, dim3((unsigned)bx), dim3(256), 0, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}
