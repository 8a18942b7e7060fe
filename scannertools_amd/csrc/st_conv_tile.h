// Internal to the convolution sources (st_conv.hip: per-tap kernels, kernel choice, C ABI; st_conv_tile_bf16x3.hip and
// st_conv_tile_f32.hip: the spatial-tile kernels, in files of their own so that their fully unrolled instances compile in parallel).
#ifndef ST_CONV_TILE_H_
#define ST_CONV_TILE_H_

#include "st_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// two floats -> packed bf16 pairs of the three terms
#ifdef __HIPCC__
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
#endif

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

// launchers of the instances (nw: 8 = eight waves, 4 = four waves x two instruction tiles, 41 = four waves x one; kh: 3 or 7)
void st_conv_tile_launch_bf16x3(st_ctx* ctx, int kh, int nw, dim3 grid, const ConvTileArgs& ta);
void st_conv_tile_launch_f32(st_ctx* ctx, int kh, int nw, dim3 grid, const ConvTileArgs& ta);
// the weights in the tile kernels' operand order
void st_conv_tile_pack_bf16x3(st_ctx* ctx, unsigned blocks, const float* w, int cout_pad, int taps, int cin, unsigned* out);
void st_conv_tile_pack_f32(st_ctx* ctx, unsigned blocks, const float* w, int cout_pad, int taps, int cin, float* out);

#endif  // ST_CONV_TILE_H_
