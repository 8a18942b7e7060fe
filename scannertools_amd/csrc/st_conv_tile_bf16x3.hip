// The spatial-tile convolution kernel of the pose network in bf16x3 arithmetic (see st_conv.hip for the stack, the per-tap kernels
// and the C ABI; cpm2_kernel.cpp:8-52 -> CaffeKernel::execute is what the stack replaces).
#include "st_conv_tile.h"

namespace {
// ---------------------------------------------------------------------------------------------------------------
// bf16x3 on a SPATIAL tile (round 4).  k_conv_nhwc_bf16x3 pulls, per 16-channel slice of one tap, 8 KB of float32
// pixels and 12 KB of split weights through the L1 for a 128 x 128 tile -- 31 B per clock and CU at full matrix rate,
// more than a CU's miss queue sustains; its matrix pipe is busy 43 % of the time (profiles/NOTES.md II 4.11).  A 7x7 layer fetches
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
}  // namespace

void st_conv_tile_launch_bf16x3(st_ctx* ctx, int kh, int nw, dim3 grid, const ConvTileArgs& ta) {
  if (kh == 7 && nw == 41) hipLaunchKernelGGL((k_conv_tile_bf16x3<7, 4, 1>), grid, dim3(256), 0, ctx->stream, ta);
  else if (kh == 7) hipLaunchKernelGGL((k_conv_tile_bf16x3<7, 4>), grid, dim3(256), 0, ctx->stream, ta);
  else if (nw == 41) hipLaunchKernelGGL((k_conv_tile_bf16x3<3, 4, 1>), grid, dim3(256), 0, ctx->stream, ta);
  else hipLaunchKernelGGL((k_conv_tile_bf16x3<3, 4>), grid, dim3(256), 0, ctx->stream, ta);
}

void st_conv_tile_pack_bf16x3(st_ctx* ctx, unsigned blocks, const float* w, int cout_pad, int taps, int cin, unsigned* out) {
  hipLaunchKernelGGL(k_pack_weights_bf16x3_tile, dim3(blocks), dim3(256), 0, ctx->stream, w, cout_pad, taps, cin, out);
}
