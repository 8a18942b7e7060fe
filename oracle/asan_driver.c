/* Sanitizer driver for the oracle (test infrastructure): compiled together with oracle.c under
 * -fsanitize=address,undefined and run on small and ragged inputs; any out-of-bounds access,
 * leak or UB aborts with a non-zero exit status.  Built and run by tests/test_oracle.py. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int num_levels; double pyr_scale; int fast_pyramids; int win_size; int num_iters; int poly_n;
  double poly_sigma; int flags; int gray_bits;
} orc_fb_params;
void orc_hist_u8c3(const uint8_t*, int, int, int, int32_t*);
void orc_gray_u8(const uint8_t*, int, int, int, uint8_t*);
void orc_fb_params_default(orc_fb_params*);
void orc_optical_flow_rgb(const uint8_t*, const uint8_t*, int, int, const orc_fb_params*, float*);
void orc_resize_linear_f32(const float*, int, int, int, float*, int, int);
void orc_gaussian_blur_f32(const float*, int, int, int, double, float*);

static uint32_t rng = 12345;
static uint8_t rnd8(void) { rng = rng * 1664525u + 1013904223u; return (uint8_t)(rng >> 24); }

int main(void) {
  static const int sizes[][2] = {{1, 1}, {1, 40}, {40, 1}, {2, 2}, {5, 7}, {31, 33}, {48, 64}, {97, 131}, {130, 70}};
  orc_fb_params p;
  orc_fb_params_default(&p);
  for (unsigned s = 0; s < sizeof(sizes) / sizeof(sizes[0]); ++s) {
    int h = sizes[s][0], w = sizes[s][1];
    size_t n = (size_t)h * w;
    uint8_t* a = malloc(3 * n); uint8_t* b = malloc(3 * n); uint8_t* g = malloc(n);
    float* flow = malloc(sizeof(float) * 2 * n);
    for (size_t i = 0; i < 3 * n; ++i) { a[i] = rnd8(); b[i] = rnd8(); }
    int32_t hist[3 * 256];
    int bins[] = {1, 16, 17, 256};
    for (int k = 0; k < 4; ++k) {
      orc_hist_u8c3(a, h, w, bins[k], hist);
      long tot = 0;
      for (int i = 0; i < 3 * bins[k]; ++i) tot += hist[i];
      if (tot != 3 * (long)n) { fprintf(stderr, "hist sum mismatch\n"); return 2; }
    }
    orc_gray_u8(a, h, w, 15, g);
    orc_gray_u8(a, h, w, 14, g);
    orc_optical_flow_rgb(a, b, h, w, &p, flow);
    p.num_levels = 5; p.win_size = 9; p.poly_n = 7;
    orc_optical_flow_rgb(a, b, h, w, &p, flow);
    orc_fb_params_default(&p);
    /* resize up, down and to odd sizes; blur with every kernel size the path uses */
    float* f = malloc(sizeof(float) * n * 2); float* d = malloc(sizeof(float) * (size_t)(2 * h + 3) * (2 * w + 3) * 2);
    for (size_t i = 0; i < 2 * n; ++i) f[i] = (float)rnd8();
    orc_resize_linear_f32(f, h, w, 2, d, 2 * h + 3, 2 * w + 1);
    orc_resize_linear_f32(f, h, w, 1, d, (h + 1) / 2, (w + 1) / 2);
    int ks[] = {3, 5, 9, 19};
    for (int k = 0; k < 4; ++k) orc_gaussian_blur_f32(f, h, w, ks[k], k ? 0.5 * k + 0.5 : 0.0, d);
    free(a); free(b); free(g); free(flow); free(f); free(d);
  }
  printf("oracle sanitizer run ok\n");
  return 0;
}
