// The spatial-tile convolution kernel of the pose network for the float32 matrix instruction (see st_conv.hip and
// st_conv_tile_bf16x3.hip, whose structure it shares).
#include "st_conv_tile.h"

namespace {
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
}  // namespace

void st_conv_tile_launch_f32(st_ctx* ctx, int kh, int nw, dim3 grid, const ConvTileArgs& ta) {
  if (kh == 7 && nw == 8) hipLaunchKernelGGL((k_conv_tile_f32<7, 8>), grid, dim3(512), 0, ctx->stream, ta);
  else if (kh == 7 && nw == 41) hipLaunchKernelGGL((k_conv_tile_f32<7, 4, 1>), grid, dim3(256), 0, ctx->stream, ta);
  else if (kh == 7) hipLaunchKernelGGL((k_conv_tile_f32<7, 4>), grid, dim3(256), 0, ctx->stream, ta);
  else if (nw == 8) hipLaunchKernelGGL((k_conv_tile_f32<3, 8>), grid, dim3(512), 0, ctx->stream, ta);
  else if (nw == 41) hipLaunchKernelGGL((k_conv_tile_f32<3, 4, 1>), grid, dim3(256), 0, ctx->stream, ta);
  else hipLaunchKernelGGL((k_conv_tile_f32<3, 4>), grid, dim3(256), 0, ctx->stream, ta);
}

void st_conv_tile_pack_f32(st_ctx* ctx, unsigned blocks, const float* w, int cout_pad, int taps, int cin, float* out) {
  hipLaunchKernelGGL(k_pack_weights_f32_tile, dim3(blocks), dim3(256), 0, ctx->stream, w, cout_pad, taps, cin, out);
}
