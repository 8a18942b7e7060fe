// Micro-benchmark: vector-memory instruction throughput per CU for the load shapes of k_flow_iter
// (coalesced dword rows of R0, 4-B-aligned dwordx2 pairs of the bilinear R1 gather), at the
// kernel's occupancy (8 waves per CU), with an L2-resident footprint so that neither HBM nor the
// fabric is the limit.  Prints CU clocks per wave-instruction (2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

// MODE 0: dword, lane stride 4 B, 256-B aligned rows     (R0 planes)
// MODE 1: dword, lane stride 4 B, rows shifted by 4 B    (unaligned coalesced)
// MODE 2: dwordx2 at lane stride 4 B (overlapping pairs)  (R1 bilinear pair x, x+1)
// MODE 3: dwordx2, lane stride 8 B (coalesced)
// MODE 4: dwordx4, lane stride 16 B (coalesced)
// MODE 5: dwordx2 pairs with a per-lane jitter of 0..3 px (non-uniform flow)
// MODE 6/7/8: dwordx4 / dwordx2 / dword at a 20-B lane stride (interleaved 5-float pixels)
template <int MODE, int U>
__global__ __launch_bounds__(256) void k(const float* __restrict__ a, float* __restrict__ out, int row_floats, int rows, int iters) {
  float s = 0;
  const int lane = threadIdx.x;
  int r = (blockIdx.x * 7) % rows;
  const int jit = MODE == 5 ? (int)((threadIdx.x * 2654435761u >> 13) & 3) : 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float* row = a + (size_t)((r + u * 3) % rows) * row_floats;
      if (MODE == 0) s += row[lane];
      if (MODE == 1) s += row[lane + 1];
      if (MODE == 2) { f2u v = *(const f2u*)(row + lane + 1); s += v.x + v.y; }
      if (MODE == 3) { f2u v = *(const f2u*)(row + 2 * lane); s += v.x + v.y; }
      if (MODE == 4) { f4u v = *(const f4u*)(row + 4 * lane); s += v.x + v.y + v.z + v.w; }
      if (MODE == 6) { f4u v = *(const f4u*)(row + 5 * lane + 1); s += v.x + v.y + v.z + v.w; }
      if (MODE == 7) { f2u v = *(const f2u*)(row + 5 * lane + 9); s += v.x + v.y; }
      if (MODE == 8) s += row[5 * lane + 4];
      if (MODE == 5) { f2u v = *(const f2u*)(row + lane + 1 + jit); s += v.x + v.y; }
    }
    r = (r + U * 3 + 1) % rows;
  }
  if (s == 1234.5f) out[0] = s;
}

template <int MODE, int U = 16>
int run(const char* name, const float* a, float* o, int row_floats, int rows, int wg_per_cu = 2) {
  const int iters = 400, blocks = 256 * wg_per_cu;  // workgroups of 4 waves
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, U>), dim3(blocks), dim3(256), 0, 0, a, o, row_floats, rows, 20);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, U>), dim3(blocks), dim3(256), 0, 0, a, o, row_floats, rows, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double winstr = (double)blocks * 4 * iters * U;  // wave-instructions
  const double bytes_per = (MODE == 4 || MODE == 6) ? 1024 : MODE == 8 ? 256 : (MODE == 0 || MODE == 1) ? 256 : 512;
  printf("%d wg/CU, %2d loads in flight: %-44s %.3f ms  %.1f clk/wave-instr/CU  %.1f requested B/clk/CU\n", wg_per_cu, U, name, ms, ms * 1e-3 * 2.4e9 * 256 / winstr,
         winstr * bytes_per / (ms * 1e-3 * 2.4e9 * 256));
  return 0;
}

int main() {
  const int row_floats = 2048, rows = 1024;  // 8 MB: L2 resident
  float *a, *o;
  CK(hipMalloc(&a, (size_t)row_floats * rows * 4 + 4096)); CK(hipMalloc(&o, 4));
  CK(hipMemset(a, 0, (size_t)row_floats * rows * 4 + 4096));
  run<0>("dword, coalesced, aligned rows", a, o, row_floats, rows);
  run<1>("dword, coalesced, rows shifted 4 B", a, o, row_floats, rows);
  run<2>("dwordx2 at 4-B lane stride (bilinear pair)", a, o, row_floats, rows);
  run<5>("dwordx2 pair, per-lane jitter 0..3 px", a, o, row_floats, rows);
  run<3>("dwordx2, coalesced", a, o, row_floats, rows);
  run<4>("dwordx4, coalesced", a, o, row_floats, rows);
  run<6>("dwordx4 at 20-B lane stride (AoS pixel)", a, o, row_floats, rows);
  run<7>("dwordx2 at 20-B lane stride", a, o, row_floats, rows);
  run<8>("dword at 20-B lane stride", a, o, row_floats, rows);
  for (int wg = 1; wg <= 8; wg *= 2) {
    run<0>("dword, coalesced, aligned rows", a, o, row_floats, rows, wg);
    run<2>("dwordx2 at 4-B lane stride (bilinear pair)", a, o, row_floats, rows, wg);
    run<4>("dwordx4, coalesced", a, o, row_floats, rows, wg);
  }
  run<0, 32>("dword, coalesced, aligned rows", a, o, row_floats, rows, 2);
  run<2, 32>("dwordx2 at 4-B lane stride (bilinear pair)", a, o, row_floats, rows, 2);
  run<2, 48>("dwordx2 at 4-B lane stride (bilinear pair)", a, o, row_floats, rows, 2);
  return 0;
}
