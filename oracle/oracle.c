/*
 * oracle.c -- CPU restatement of the arithmetic behind scannertools' Histogram and
 * OpticalFlow ops.  TEST INFRASTRUCTURE ONLY: nothing under scannertools_amd/ may
 * include, link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / CPU baseline.
 *
 * What it restates
 * ----------------
 * The reference ops are thin wrappers over OpenCV (un-vendored, un-pinned: "OpenCV >= 3.4.0",
 * /root/reference/scannertools/README.md:10; the code is OpenCV-3.4/4.x era):
 *   Histogram   scannertools_cpp/imgproc/histogram_kernel_cpu.cpp:33-42
 *               cv::calcHist(8U, 1 channel, BINS uniform bins on [0,256)) x3 + convertTo(CV_32S)
 *   OpticalFlow scannertools_cpp/imgproc/optical_flow_kernel_cpu.cpp:15-16,36-41
 *               cv::cvtColor(COLOR_BGR2GRAY) x2 (applied to RGB data: reference quirk) and
 *               cv::FarnebackOpticalFlow::create(3, 0.5, false, 15, 3, 5, 1.2, 0)->calc(g0, g1, flow)
 * OpenCV is not present in /root/reference nor in the build container, so each function
 * below restates the published OpenCV algorithm (modules/imgproc/src/{histogram,color_rgb,
 * smooth,filter,resize}.*, modules/video/src/optflowgf.cpp), keeping its operand order,
 * accumulator types (float vs double), border rules and rounding, and is anchored on the
 * reference call sites above.  Compile with -ffp-contract=off so that float expressions
 * are evaluated exactly as written (scalar, non-FMA OpenCV build).
 *
 * PARITY STATUS
 *   Histogram : pinned by definition (integer; bin = floor(v*bins/256)), cross-checked
 *               against numpy.bincount in tests/.
 *   OpticalFlow: PARITY UNPINNED against real OpenCV output -- the reference's own tests
 *               only assert dtype/shape (scannertools/tests/test_all.py:162-177) and no
 *               OpenCV build exists here to emit golden vectors.  The oracle is pinned
 *               only by analytic known-answer tests (polynomial expansion of exact
 *               quadratics, recovered integer translations, zero-flow identities) and
 *               by an independent float64 derivation of the published algorithm
 *               (tests/ref_farneback_np.py; stages and end-to-end flow agree to float32
 *               rounding).
 * The later sections restate the ops either side of the path (each with its own header):
 *   DrawFlow    : pinned by golden vectors produced by importing the reference's vis.py.
 *   Blur        : pinned by the reference source itself (blur_kernel_cpu.cpp spells out the
 *                 integer arithmetic; no OpenCV involved).
 *   FlowHistogram, Resize, ConvertColor: PARITY UNPINNED against real OpenCV output
 *                 (cartToPolar/calcHist, resize, cvtColor restated; known answers only).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
/* cvRound: round half to even (SSE cvtsd2si / lrint under the default rounding mode). */
static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) { return (int)floorf(v); }

/* ------------------------------------------------------------------------------------------
 * A1  cv::calcHist, 8U uniform path (call site histogram_kernel_cpu.cpp:36-39).
 * One channel j of an interleaved (h,w,3) U8 frame; tab[v] = floor(v * bins/256).
 * out: 3*bins int32, channel-major (histogram_kernel_cpu.cpp:40-41 writes channel j at
 * output_buf + j*BINS*sizeof(int)).
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_hist_u8c3(const uint8_t* frame, int h, int w, int bins, int32_t* out) {
  int tab[256];
  double a = (double)bins / 256.0;
  for (int v = 0; v < 256; ++v) tab[v] = (int)floor(v * a);
  memset(out, 0, sizeof(int32_t) * 3 * (size_t)bins);
  size_t n = (size_t)h * w;
  for (size_t i = 0; i < n; ++i) {
    out[0 * bins + tab[frame[3 * i + 0]]]++;
    out[1 * bins + tab[frame[3 * i + 1]]]++;
    out[2 * bins + tab[frame[3 * i + 2]]]++;
  }
}

/* ------------------------------------------------------------------------------------------
 * A2  cv::cvtColor(COLOR_BGR2GRAY), 8-bit (call sites optical_flow_kernel_cpu.cpp:38-39).
 * blueIdx = 0: byte 0 gets the B weight, byte 2 the R weight.  The reference feeds RGB
 * frames, so byte 0 is really R: gray = 0.114 R + 0.587 G + 0.299 B.  Keep the quirk.
 * bits = 15 (OpenCV 4.x / late 3.4: 3735, 19235, 9798) or 14 (older: 1868, 9617, 4899).
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_gray_u8(const uint8_t* src3, int h, int w, int bits, uint8_t* gray) {
  int cb, cg, cr;
  if (bits == 14) { cb = 1868; cg = 9617; cr = 4899; } else { bits = 15; cb = 3735; cg = 19235; cr = 9798; }
  int rnd = 1 << (bits - 1);
  size_t n = (size_t)h * w;
  for (size_t i = 0; i < n; ++i)
    gray[i] = (uint8_t)((src3[3 * i] * cb + src3[3 * i + 1] * cg + src3[3 * i + 2] * cr + rnd) >> bits);
}

/* ------------------------------------------------------------------------------------------
 * cv::getGaussianKernel(n, sigma, CV_32F): fixed table for n=3, sigma<=0; else exp() in
 * double, stored to float, normalised by the double sum of the floats.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_gaussian_kernel(int n, double sigma, float* k) {
  if (n == 3 && sigma <= 0) { k[0] = 0.25f; k[1] = 0.5f; k[2] = 0.25f; return; }
  if (n == 1 && sigma <= 0) { k[0] = 1.f; return; }
  if (n == 5 && sigma <= 0) { k[0] = 0.0625f; k[1] = 0.25f; k[2] = 0.375f; k[3] = 0.25f; k[4] = 0.0625f; return; }
  if (n == 7 && sigma <= 0) {
    static const float t[7] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
    memcpy(k, t, sizeof(t)); return;
  }
  double sx = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  double scale2x = -0.5 / (sx * sx);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    double x = i - (n - 1) * 0.5;
    k[i] = (float)exp(scale2x * x * x);
    sum += k[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) k[i] = (float)(k[i] * sum);
}

static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p; else p = 2 * (len - 1) - p;
  }
  return p;
}

/* ------------------------------------------------------------------------------------------
 * cv::GaussianBlur on CV_32F = sepFilter2D(row filter, then column filter), float
 * intermediate, BORDER_REFLECT_101.  Operand order per OpenCV's filter engine:
 *   row,  ks<=5 (SymmRowSmallFilter): c*k0 + (l1+r1)*k1 [+ (l2+r2)*k2]
 *   row,  ks>5  (RowFilter)         : left-to-right accumulation over the taps
 *   col,  ks==3 (SymmColumnSmallFilter): (top+bottom)*k1 + c*k0
 *   col,  ks>3  (SymmColumnFilter)  : c*k0, then += k[j]*(S[+j] + S[-j]) for j=1..
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_gaussian_blur_f32(const float* src, int h, int w, int ks, double sigma, float* dst) {
  float kbuf[64];
  orc_gaussian_kernel(ks, sigma, kbuf);
  int r = ks / 2;
  const float* kc = kbuf + r;
  float* tmp = (float*)malloc(sizeof(float) * (size_t)h * w);
  for (int y = 0; y < h; ++y) {
    const float* S = src + (size_t)y * w;
    float* D = tmp + (size_t)y * w;
    for (int x = 0; x < w; ++x) {
      float s;
      if (ks == 3) {
        s = S[x] * kc[0] + (S[reflect101(x - 1, w)] + S[reflect101(x + 1, w)]) * kc[1];
      } else if (ks == 5) {
        s = S[x] * kc[0] + (S[reflect101(x - 1, w)] + S[reflect101(x + 1, w)]) * kc[1] +
            (S[reflect101(x - 2, w)] + S[reflect101(x + 2, w)]) * kc[2];
      } else {
        s = kbuf[0] * S[reflect101(x - r, w)];
        for (int k = 1; k < ks; ++k) s += kbuf[k] * S[reflect101(x - r + k, w)];
      }
      D[x] = s;
    }
  }
  for (int y = 0; y < h; ++y) {
    float* D = dst + (size_t)y * w;
    if (ks == 3) {
      const float* S0 = tmp + (size_t)reflect101(y - 1, h) * w;
      const float* S1 = tmp + (size_t)y * w;
      const float* S2 = tmp + (size_t)reflect101(y + 1, h) * w;
      for (int x = 0; x < w; ++x) D[x] = (S0[x] + S2[x]) * kc[1] + S1[x] * kc[0];
    } else {
      const float* Sc = tmp + (size_t)y * w;
      for (int x = 0; x < w; ++x) D[x] = kc[0] * Sc[x];
      for (int k = 1; k <= r; ++k) {
        const float* Sp = tmp + (size_t)reflect101(y + k, h) * w;
        const float* Sm = tmp + (size_t)reflect101(y - k, h) * w;
        for (int x = 0; x < w; ++x) D[x] += kc[k] * (Sp[x] + Sm[x]);
      }
    }
  }
  free(tmp);
}

/* ------------------------------------------------------------------------------------------
 * cv::resize(..., INTER_LINEAR) for CV_32FC(cn): half-pixel centres, horizontal pass then
 * vertical pass in float; exact-2x decimation is rerouted by OpenCV to the INTER_AREA fast
 * path ((a+b)+(c+d))*0.25; equal sizes are a plain copy.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_resize_linear_f32(const float* src, int sh, int sw, int cn, float* dst, int dh, int dw) {
  if (sh == dh && sw == dw) { memcpy(dst, src, sizeof(float) * (size_t)sh * sw * cn); return; }
  double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
  double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
  if (sw == 2 * dw && sh == 2 * dh) {
    for (int y = 0; y < dh; ++y)
      for (int x = 0; x < dw; ++x)
        for (int c = 0; c < cn; ++c) {
          const float* S0 = src + ((size_t)(2 * y) * sw + 2 * x) * cn + c;
          const float* S1 = S0 + (size_t)sw * cn;
          dst[((size_t)y * dw + x) * cn + c] = ((S0[0] + S0[cn]) + (S1[0] + S1[cn])) * 0.25f;
        }
    return;
  }
  int* xofs = (int*)malloc(sizeof(int) * dw);
  float* xa = (float*)malloc(sizeof(float) * dw);
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx; xa[dx] = fx;
  }
  float* rows[2];
  rows[0] = (float*)malloc(sizeof(float) * (size_t)dw * cn);
  rows[1] = (float*)malloc(sizeof(float) * (size_t)dw * cn);
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor_f(fy);
    fy -= sy;
    float b0 = 1.f - fy, b1 = fy;
    for (int k = 0; k < 2; ++k) {
      int yy = imin(imax(sy + k, 0), sh - 1);
      const float* S = src + (size_t)yy * sw * cn;
      for (int dx = 0; dx < dw; ++dx) {
        int sx = xofs[dx];
        float a1 = xa[dx], a0 = 1.f - a1;
        for (int c = 0; c < cn; ++c) {
          if (sx + 1 < sw) rows[k][dx * cn + c] = S[sx * cn + c] * a0 + S[(sx + 1) * cn + c] * a1;
          else rows[k][dx * cn + c] = S[sx * cn + c] * 1.f;
        }
      }
    }
    float* D = dst + (size_t)dy * dw * cn;
    for (int i = 0; i < dw * cn; ++i) D[i] = rows[0][i] * b0 + rows[1][i] * b1;
  }
  free(rows[0]); free(rows[1]); free(xofs); free(xa);
}

/* ------------------------------------------------------------------------------------------
 * FarnebackPrepareGaussian (optflowgf.cpp): weights g, x*g, x^2*g for x in [-n,n] and the
 * four entries of inv(G) that the expansion uses.
 * g/xg/xxg point at the centre tap (index 0), valid for [-n, n].
 * ---------------------------------------------------------------------------------------- */
static void invert6(double A[6][6], double inv[6][6]) {
  double a[6][12];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) { a[i][j] = A[i][j]; a[i][6 + j] = (i == j); }
  for (int c = 0; c < 6; ++c) {
    int p = c;
    for (int r = c + 1; r < 6; ++r) if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
    if (p != c) for (int j = 0; j < 12; ++j) { double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
    double d = 1. / a[c][c];
    for (int j = 0; j < 12; ++j) a[c][j] *= d;
    for (int r = 0; r < 6; ++r) if (r != c) {
      double f = a[r][c];
      if (f != 0) for (int j = 0; j < 12; ++j) a[r][j] -= f * a[c][j];
    }
  }
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) inv[i][j] = a[i][6 + j];
}

ORC_API void orc_poly_prepare(int n, double sigma, float* g, float* xg, float* xxg, double* ig /* ig11, ig03, ig33, ig55 */) {
  if (sigma < 1.1920929e-07) sigma = n * 0.3;
  double s = 0.;
  for (int x = -n; x <= n; ++x) {
    g[x] = (float)exp(-x * x / (2 * sigma * sigma));
    s += g[x];
  }
  s = 1. / s;
  for (int x = -n; x <= n; ++x) {
    g[x] = (float)(g[x] * s);
    xg[x] = (float)(x * g[x]);
    xxg[x] = (float)(x * x * g[x]);
  }
  double G[6][6];
  memset(G, 0, sizeof(G));
  for (int y = -n; y <= n; ++y)
    for (int x = -n; x <= n; ++x) {
      G[0][0] += g[y] * g[x];
      G[1][1] += g[y] * g[x] * x * x;
      G[3][3] += g[y] * g[x] * x * x * x * x;
      G[5][5] += g[y] * g[x] * x * x * y * y;
    }
  G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
  G[4][4] = G[3][3];
  G[3][4] = G[4][3] = G[5][5];
  double inv[6][6];
  invert6(G, inv);
  ig[0] = inv[1][1]; ig[1] = inv[0][3]; ig[2] = inv[3][3]; ig[3] = inv[5][5];
}

/* ------------------------------------------------------------------------------------------
 * A4  FarnebackPolyExp(I, R, n, sigma).  Vertical pass in float with rows clamped,
 * horizontal pass with replicated columns; b1/b4 accumulate double products, b2,b3,b5,b6
 * accumulate float products widened to double (the expression types of the original).
 * R: (h,w,5) interleaved.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_polyexp(const float* I, int h, int w, int n, double sigma, float* R) {
  float kbuf[3 * 33];
  float* g = kbuf + n;
  float* xg = g + 2 * n + 1;
  float* xxg = xg + 2 * n + 1;
  double ig[4];
  orc_poly_prepare(n, sigma, g, xg, xxg, ig);
  double ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];
  float* rowbuf = (float*)malloc(sizeof(float) * (size_t)(w + 2 * n) * 3);
  float* row = rowbuf + n * 3;
  for (int y = 0; y < h; ++y) {
    float g0 = g[0], g1, g2;
    const float* srow0 = I + (size_t)y * w;
    const float* srow1;
    float* drow = R + (size_t)y * w * 5;
    for (int x = 0; x < w; ++x) {
      row[x * 3] = srow0[x] * g0;
      row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
    }
    for (int k = 1; k <= n; ++k) {
      g0 = g[k]; g1 = xg[k]; g2 = xxg[k];
      srow0 = I + (size_t)imax(y - k, 0) * w;
      srow1 = I + (size_t)imin(y + k, h - 1) * w;
      for (int x = 0; x < w; ++x) {
        float p = srow0[x] + srow1[x];
        float t0 = row[x * 3] + g0 * p;
        float t1 = row[x * 3 + 2] + g2 * p;
        p = srow1[x] - srow0[x];
        float t2 = row[x * 3 + 1] + g1 * p;
        row[x * 3] = t0;
        row[x * 3 + 1] = t2;
        row[x * 3 + 2] = t1;
      }
    }
    for (int x = 0; x < n * 3; ++x) {
      row[-1 - x] = row[2 - x];
      row[w * 3 + x] = row[w * 3 + x - 3];
    }
    for (int x = 0; x < w; ++x) {
      g0 = g[0];
      double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0, b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
      for (int k = 1; k <= n; ++k) {
        double tg = row[(x + k) * 3] + row[(x - k) * 3];
        g0 = g[k];
        b1 += tg * g0;
        b4 += tg * xxg[k];
        b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
        b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
        b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
        b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
      }
      drow[x * 5 + 1] = (float)(b2 * ig11);
      drow[x * 5] = (float)(b3 * ig11);
      drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
      drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
      drow[x * 5 + 4] = (float)(b6 * ig55);
    }
  }
  free(rowbuf);
}

/* ------------------------------------------------------------------------------------------
 * A5  FarnebackUpdateMatrices(R0, R1, flow, M, y0, y1).  R*, M: (h,w,5); flow: (h,w,2).
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_update_matrices(const float* R0a, const float* R1a, const float* flowa, float* Ma,
                                 int h, int w, int y0, int y1) {
  enum { BORDER = 5 };
  static const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
  size_t step1 = (size_t)w * 5;
  for (int y = y0; y < y1; ++y) {
    const float* flow = flowa + (size_t)y * w * 2;
    const float* R0 = R0a + (size_t)y * w * 5;
    float* M = Ma + (size_t)y * w * 5;
    for (int x = 0; x < w; ++x) {
      float dx = flow[x * 2], dy = flow[x * 2 + 1];
      float fx = x + dx, fy = y + dy;
      int x1 = cv_floor_f(fx), y1i = cv_floor_f(fy);
      float r2, r3, r4, r5, r6;
      fx -= x1; fy -= y1i;
      if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1i < (unsigned)(h - 1)) {
        const float* ptr = R1a + (size_t)y1i * step1 + (size_t)x1 * 5;
        float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
        r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
        r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
        r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
        r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
        r4 = (R0[x * 5 + 2] + r4) * 0.5f;
        r5 = (R0[x * 5 + 3] + r5) * 0.5f;
        r6 = (R0[x * 5 + 4] + r6) * 0.25f;
      } else {
        r2 = r3 = 0.f;
        r4 = R0[x * 5 + 2];
        r5 = R0[x * 5 + 3];
        r6 = R0[x * 5 + 4] * 0.5f;
      }
      r2 = (R0[x * 5] - r2) * 0.5f;
      r3 = (R0[x * 5 + 1] - r3) * 0.5f;
      r2 += r4 * dy + r6 * dx;
      r3 += r6 * dy + r5 * dx;
      if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) ||
          (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
        float scale = (x < BORDER ? border[x] : 1.f) * (x >= w - BORDER ? border[w - x - 1] : 1.f) *
                      (y < BORDER ? border[y] : 1.f) * (y >= h - BORDER ? border[h - y - 1] : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
      }
      M[x * 5] = r4 * r4 + r6 * r6;
      M[x * 5 + 1] = (r4 + r5) * r6;
      M[x * 5 + 2] = r5 * r5 + r6 * r6;
      M[x * 5 + 3] = r4 * r2 + r6 * r3;
      M[x * 5 + 4] = r6 * r2 + r5 * r3;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * A6  FarnebackUpdateFlow_Blur(R0, R1, flow, M, block_size, update_matrices): box filter of M
 * by double running sums (vertical: += float(M[y+m] - M[y-m-1]); horizontal: running window
 * over the replicated vsum row), 2x2 solve, then striped in-place UpdateMatrices.
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_update_flow_blur(const float* R0, const float* R1, float* flowa, float* Ma,
                                  int h, int w, int block_size, int update_matrices) {
  int m = block_size / 2;
  int y0 = 0, y1;
  int min_update_stripe = imax((1 << 10) / w, block_size);
  double scale = 1. / (block_size * block_size);
  double* vbuf = (double*)malloc(sizeof(double) * (size_t)(w + m * 2 + 2) * 5);
  double* vsum = vbuf + (m + 1) * 5;
  const float* srow0 = Ma;
  for (int x = 0; x < w * 5; ++x) vsum[x] = srow0[x] * (m + 2);
  for (int y = 1; y < m; ++y) {
    srow0 = Ma + (size_t)imin(y, h - 1) * w * 5;
    for (int x = 0; x < w * 5; ++x) vsum[x] += srow0[x];
  }
  for (int y = 0; y < h; ++y) {
    double g11, g12, g22, h1, h2;
    float* flow = flowa + (size_t)y * w * 2;
    srow0 = Ma + (size_t)imax(y - m - 1, 0) * w * 5;
    const float* srow1 = Ma + (size_t)imin(y + m, h - 1) * w * 5;
    for (int x = 0; x < w * 5; ++x) vsum[x] += srow1[x] - srow0[x];
    for (int x = 0; x < (m + 1) * 5; ++x) {
      vsum[-1 - x] = vsum[4 - x];
      vsum[w * 5 + x] = vsum[w * 5 + x - 5];
    }
    g11 = vsum[0] * (m + 2);
    g12 = vsum[1] * (m + 2);
    g22 = vsum[2] * (m + 2);
    h1 = vsum[3] * (m + 2);
    h2 = vsum[4] * (m + 2);
    for (int x = 1; x < m; ++x) {
      g11 += vsum[x * 5];
      g12 += vsum[x * 5 + 1];
      g22 += vsum[x * 5 + 2];
      h1 += vsum[x * 5 + 3];
      h2 += vsum[x * 5 + 4];
    }
    for (int x = 0; x < w; ++x) {
      g11 += vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5];
      g12 += vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4];
      g22 += vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3];
      h1 += vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2];
      h2 += vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1];
      double g11_ = g11 * scale, g12_ = g12 * scale, g22_ = g22 * scale, h1_ = h1 * scale, h2_ = h2 * scale;
      double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
      flow[x * 2] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
      flow[x * 2 + 1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
    }
    y1 = y == h - 1 ? h : y - block_size;
    if (update_matrices && (y1 == h || y1 >= y0 + min_update_stripe)) {
      orc_update_matrices(R0, R1, flowa, Ma, h, w, y0, y1);
      y0 = y1;
    }
  }
  free(vbuf);
}

/* Parameters of cv::FarnebackOpticalFlow::create(numLevels, pyrScale, fastPyramids, winSize,
 * numIters, polyN, polySigma, flags).  Reference: (3, 0.5, false, 15, 3, 5, 1.2, 0)
 * optical_flow_kernel_cpu.cpp:16.  Only flags == 0 and fastPyramids == false are restated. */
typedef struct orc_fb_params {
  int num_levels;
  double pyr_scale;
  int fast_pyramids;
  int win_size;
  int num_iters;
  int poly_n;
  double poly_sigma;
  int flags;
  int gray_bits;
} orc_fb_params;

ORC_API void orc_fb_params_default(orc_fb_params* p) {
  p->num_levels = 3; p->pyr_scale = 0.5; p->fast_pyramids = 0; p->win_size = 15; p->num_iters = 3;
  p->poly_n = 5; p->poly_sigma = 1.2; p->flags = 0; p->gray_bits = 15;
}

/* Number of pyramid levels actually processed is levels+1 (k = levels .. 0). */
ORC_API int orc_fb_levels(int h, int w, const orc_fb_params* p) {
  int k; double scale = 1;
  for (k = 0; k < p->num_levels; ++k) {
    scale *= p->pyr_scale;
    if (w * scale < 32 || h * scale < 32) break;
  }
  return k;
}

ORC_API void orc_fb_level_geom(int h, int w, const orc_fb_params* p, int k, int* lh, int* lw, double* sigma, int* ksize) {
  double scale = 1;
  for (int i = 0; i < k; ++i) scale *= p->pyr_scale;
  double sg = (1. / scale - 1) * 0.5;
  int sz = cv_round(sg * 5) | 1;
  sz = imax(sz, 3);
  *lw = cv_round(w * scale); *lh = cv_round(h * scale); *sigma = sg; *ksize = sz;
}

/* One pyramid image: convertTo(F32) -> GaussianBlur(full res) -> resize to level size. */
ORC_API void orc_fb_pyr_image(const uint8_t* gray, int h, int w, const orc_fb_params* p, int k, float* I) {
  int lh, lw, ks; double sigma;
  orc_fb_level_geom(h, w, p, k, &lh, &lw, &sigma, &ks);
  size_t n = (size_t)h * w;
  float* f = (float*)malloc(sizeof(float) * n);
  float* b = (float*)malloc(sizeof(float) * n);
  for (size_t i = 0; i < n; ++i) f[i] = (float)gray[i];
  orc_gaussian_blur_f32(f, h, w, ks, sigma, b);
  orc_resize_linear_f32(b, h, w, 1, I, lh, lw);
  free(f); free(b);
}

/* ------------------------------------------------------------------------------------------
 * A3  FarnebackOpticalFlowImpl::calc(prev, next, flow) for flags = 0.
 * flow: (h,w,2) F32 interleaved (u,v): next(x+u, y+v) ~ prev(x,y).
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_farneback(const uint8_t* prev, const uint8_t* next, int h, int w, const orc_fb_params* p, float* flow_out) {
  const uint8_t* img[2] = {prev, next};
  int levels = orc_fb_levels(h, w, p);
  float* prev_flow = NULL; int ph = 0, pw = 0;
  for (int k = levels; k >= 0; --k) {
    int lh, lw, ks; double sigma;
    orc_fb_level_geom(h, w, p, k, &lh, &lw, &sigma, &ks);
    size_t np = (size_t)lh * lw;
    float* flow = k > 0 ? (float*)malloc(sizeof(float) * np * 2) : flow_out;
    if (!prev_flow) {
      memset(flow, 0, sizeof(float) * np * 2);
    } else {
      orc_resize_linear_f32(prev_flow, ph, pw, 2, flow, lh, lw);
      float mul = (float)(1. / p->pyr_scale);
      for (size_t i = 0; i < np * 2; ++i) flow[i] = flow[i] * mul;
    }
    float* R[2];
    float* I = (float*)malloc(sizeof(float) * np);
    for (int i = 0; i < 2; ++i) {
      R[i] = (float*)malloc(sizeof(float) * np * 5);
      orc_fb_pyr_image(img[i], h, w, p, k, I);
      orc_polyexp(I, lh, lw, p->poly_n, p->poly_sigma, R[i]);
    }
    free(I);
    float* M = (float*)malloc(sizeof(float) * np * 5);
    orc_update_matrices(R[0], R[1], flow, M, lh, lw, 0, lh);
    for (int i = 0; i < p->num_iters; ++i)
      orc_update_flow_blur(R[0], R[1], flow, M, lh, lw, p->win_size, i < p->num_iters - 1);
    free(M); free(R[0]); free(R[1]);
    if (prev_flow) free(prev_flow);
    prev_flow = flow; ph = lh; pw = lw;
  }
}

/* The op: OpticalFlowKernelCPU::execute (optical_flow_kernel_cpu.cpp:36-41):
 * gray(frame0), gray(frame1), calc(gray0, gray1, flow). */
ORC_API void orc_optical_flow_rgb(const uint8_t* frame0, const uint8_t* frame1, int h, int w,
                                  const orc_fb_params* p, float* flow_out) {
  size_t n = (size_t)h * w;
  uint8_t* g0 = (uint8_t*)malloc(n);
  uint8_t* g1 = (uint8_t*)malloc(n);
  orc_gray_u8(frame0, h, w, p->gray_bits, g0);
  orc_gray_u8(frame1, h, w, p->gray_bits, g1);
  orc_farneback(g0, g1, h, w, p, flow_out);
  free(g0); free(g1);
}

/* ------------------------------------------------------------------------------------------
 * Flow consumers.
 *
 * FlowHistogram -- FlowHistogramKernelCPU::execute,
 * /root/reference/scannertools/scannertools/old/cpp_ops/flow_histogram_kernel_cpu.cpp:26-57:
 *   split(flow) -> cv::cartToPolar(x, y, mag, deg, angleInDegrees=true)
 *   cv::calcHist(mag, 64 uniform bins on [0,64)), cv::calcHist(deg, 64 uniform bins on [0,360))
 *   each converted to CV_32S; output = 2 x 64 int32 (magnitude row first).
 * OpenCV (un-vendored) arithmetic restated, scalar non-FMA build:
 *   cartToPolar 32F (core/src/mathfuncs.cpp): hal::magnitude32f = sqrt(x*x + y*y) in float;
 *     hal::fastAtan32f = atan_f32(y, x) * 1.f, the 7th-order odd polynomial of
 *     core/src/mathfuncs_core.simd.hpp (max error ~0.3 deg), result in [0, 360].
 *   calcHist 32F uniform (imgproc/src/histogram.cpp calcHist_): idx = cvFloor(double(v)*a + b)
 *     with a = bins/(hi-lo), b = -a*lo in double; counted iff 0 <= idx < bins (v == hi, NaN
 *     and negative values are dropped).
 * PARITY UNPINNED against real OpenCV output (no OpenCV here); pinned by known-answer tests
 * (axis-aligned vectors -> exact angles/magnitudes) in tests/.
 * ------------------------------------------------------------------------------------------ */
#include <float.h>
#define ORC_CV_PI 3.1415926535897932384626433832795 /* CV_PI */

static inline float orc_fast_atan2_deg(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / ORC_CV_PI);
  static const float p3 = -0.3258083974640975f * (float)(180 / ORC_CV_PI);
  static const float p5 = 0.1555786518463281f * (float)(180 / ORC_CV_PI);
  static const float p7 = -0.04432655554792128f * (float)(180 / ORC_CV_PI);
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

static inline int orc_cv_floor(double v) {
  /* cvFloor: NaN / out-of-range follow cvtsd2si (INT_MIN), which the range test rejects */
  if (!(v > -2147483648.0 && v < 2147483648.0)) return INT32_MIN;
  int i = (int)v;
  return i - (v < i);
}

ORC_API void orc_cart_to_polar_deg(const float* flow, size_t npx, float* mag, float* deg) {
  for (size_t i = 0; i < npx; ++i) {
    float x = flow[2 * i], y = flow[2 * i + 1];
    mag[i] = sqrtf(x * x + y * y);
    deg[i] = orc_fast_atan2_deg(y, x) * 1.f;
  }
}

ORC_API void orc_flow_hist(const float* flow, int h, int w, int32_t* out /* 2 x 64 */) {
  const int bins = 64;
  const double a_mag = bins / (64.0 - 0.0), b_mag = -a_mag * 0.0;
  const double a_deg = bins / (360.0 - 0.0), b_deg = -a_deg * 0.0;
  memset(out, 0, sizeof(int32_t) * 2 * bins);
  size_t npx = (size_t)h * w;
  for (size_t i = 0; i < npx; ++i) {
    float x = flow[2 * i], y = flow[2 * i + 1];
    float mag = sqrtf(x * x + y * y);
    float deg = orc_fast_atan2_deg(y, x) * 1.f;
    int im = orc_cv_floor((double)mag * a_mag + b_mag);
    int id = orc_cv_floor((double)deg * a_deg + b_deg);
    if ((unsigned)im < (unsigned)bins) out[im]++;
    if ((unsigned)id < (unsigned)bins) out[bins + id]++;
  }
}

/* DrawFlow -- /root/reference/scannertools/scannertools/vis.py:8-12 (numpy, float32):
 *   v = (fx + fy) / 2; m = max(v) over the frame (NaN propagates); out = hstack(frame,
 *   uint8(min(v / m, 1) * 255) replicated to 3 channels).  The float32 -> uint8 cast is numpy's C
 *   cast: truncate toward zero to int32 (NaN / out of range -> INT32_MIN), keep the low byte.
 * Pinned by tests/golden/draw_flow_golden.npz (produced by importing vis.py). */
ORC_API void orc_draw_flow(const uint8_t* frame, const float* flow, int h, int w, uint8_t* out /* h x 2w x 3 */) {
  size_t npx = (size_t)h * w;
  float m = -INFINITY;
  int nan = 0;
  for (size_t i = 0; i < npx; ++i) {
    float v = (flow[2 * i] + flow[2 * i + 1]) / 2.f;
    if (v != v) nan = 1;
    if (v > m) m = v;
  }
  if (nan) m = NAN;
  for (int y = 0; y < h; ++y) {
    uint8_t* o = out + (size_t)y * 2 * w * 3;
    memcpy(o, frame + (size_t)y * w * 3, (size_t)w * 3);
    for (int x = 0; x < w; ++x) {
      size_t i = (size_t)y * w + x;
      float v = (flow[2 * i] + flow[2 * i + 1]) / 2.f;
      float q = v / m;
      if (q > 1.0f) q = 1.0f;  /* np.clip(., None, 1): NaN stays NaN */
      q = q * 255.f;
      int32_t t = (q > -2147483904.f && q < 2147483648.f) ? (int32_t)q : INT32_MIN;
      uint8_t b = (uint8_t)(t & 0xff);
      o[(w + x) * 3 + 0] = b; o[(w + x) * 3 + 1] = b; o[(w + x) * 3 + 2] = b;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * Blur op -- BlurKernel::execute,
 * /root/reference/scannertools/scannertools_cpp/imgproc/blur_kernel_cpu.cpp:36-81.  The reference
 * implements the filter itself (no OpenCV): a k x k box sum over the window
 * [-filter_left, +filter_right] with filter_left = ceil(k/2.0) - 1, filter_right = k/2, unsigned
 * integer division by (filter_left + filter_right + 1)^2, 3 interleaved U8 channels; `sigma` is
 * parsed and never used.  Only interior pixels (y in [left, h-right), x in [left, w-right)) are
 * written; the reference leaves the border of its freshly allocated frame uninitialised -- this
 * restatement (and the HIP kernel) writes 0 there.  Pinned by definition: the arithmetic is spelled
 * out in the reference source and is integer.
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_box_blur_u8c3(const uint8_t* src, int h, int w, int kernel_size, uint8_t* dst) {
  const int left = (int)ceil(kernel_size / 2.0) - 1, right = kernel_size / 2;
  const unsigned div = (unsigned)((right + left + 1) * (right + left + 1));
  memset(dst, 0, (size_t)h * w * 3);
  for (int y = left; y < h - right; ++y)
    for (int x = left; x < w - right; ++x)
      for (int c = 0; c < 3; ++c) {
        uint32_t value = 0;
        for (int ry = -left; ry < right + 1; ++ry)
          for (int rx = -left; rx < right + 1; ++rx) value += src[((size_t)(y + ry) * w + (x + rx)) * 3 + c];
        dst[((size_t)y * w + x) * 3 + c] = (uint8_t)(value / div);
      }
}

/* ------------------------------------------------------------------------------------------
 * Resize op -- ResizeKernel::execute,
 * /root/reference/scannertools/scannertools_cpp/imgproc/resize_kernel.cpp:38-88: target size from
 * ResizeArgs (width/height, preserve_aspect, min), then cv::resize(img, out, Size(w, h), 0, 0,
 * interpolation) on the U8 frame (default INTER_LINEAR; the legacy flow-histogram pipeline resizes
 * to 426 x 240, old/histograms.py:64-68).
 * OpenCV (un-vendored) arithmetic restated for CV_8UC(cn), imgproc/src/resize.cpp:
 *   INTER_LINEAR : half-pixel centres fx = (float)((dx+0.5)*scale - 0.5); 11-bit fixed-point
 *                  weights saturate_cast<short>(w * 2048) (round half to even); horizontal pass
 *                  S[sx]*a0 + S[sx+cn]*a1 in int (single tap * 2048 at the right edge);
 *                  vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2
 *                  (VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>); source rows clipped
 *                  to [0, h-1].  Exact 2x2 decimation is rerouted to INTER_AREA:
 *                  (S00 + S01 + S10 + S11 + 2) >> 2 (ResizeAreaFastVec).
 *   INTER_NEAREST: sx = min(floor(dx * scale_x), sw-1), sy likewise (resizeNN).
 *   INTER_CUBIC / INTER_AREA: see orc_resize_cubic_u8 / orc_resize_area_u8 below.
 * Equal sizes are a plain copy.  PARITY UNPINNED against real OpenCV output (integer arithmetic,
 * pinned only by identities: constant images, exact 2x means, separability).
 * ------------------------------------------------------------------------------------------ */
static inline short orc_sat_short_round(float v) {
  long r = lrintf(v);  /* cvRound: round half to even (default FP environment) */
  return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

ORC_API void orc_resize_target(int fw, int fh, int width, int height, int min_flag, int preserve_aspect, int* tw, int* th) {
  /* resize_kernel.cpp:44-62 */
  int target_width = width, target_height = height;
  if (preserve_aspect) {
    if (target_width == 0) target_width = fw * target_height / fh;
    else target_height = fh * target_width / fw;
  }
  if (min_flag) {
    if (fw <= target_width && fh <= target_height) { target_width = fw; target_height = fh; }
  }
  *tw = target_width; *th = target_height;
}

/* cv::interpolateCubic (imgproc/src/resize.cpp), A = -0.75 */
static void orc_cubic_coeffs(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

static inline uint8_t orc_sat_u8_int(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static inline uint8_t orc_sat_u8_float(float v) { long r = lrintf(v); return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r)); }

/* INTER_CUBIC, 8-bit: 4 x 4 taps, 11-bit fixed-point weights, columns outside the row replicate the
 * edge pixel (HResizeCubic), rows clipped to [0, h-1], result (v + 2^21) >> 22 saturated
 * (VResizeCubic with FixedPtCast<int, uchar, 22>). */
static void orc_resize_cubic_u8(const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw,
                                double scale_x, double scale_y) {
  int* xofs = (int*)malloc(sizeof(int) * dw);
  short* ialpha = (short*)malloc(sizeof(short) * 4 * dw);
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    float c[4];
    orc_cubic_coeffs(fx, c);
    xofs[dx] = sx;
    for (int k = 0; k < 4; ++k) ialpha[4 * dx + k] = orc_sat_short_round(c[k] * 2048);
  }
  int* rows[4];
  for (int k = 0; k < 4; ++k) rows[k] = (int*)malloc(sizeof(int) * (size_t)dw * cn);
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor_f(fy);
    fy -= sy;
    float cb[4];
    orc_cubic_coeffs(fy, cb);
    short beta[4];
    for (int k = 0; k < 4; ++k) beta[k] = orc_sat_short_round(cb[k] * 2048);
    for (int k = 0; k < 4; ++k) {
      const int yy = imin(imax(sy - 1 + k, 0), sh - 1);
      const uint8_t* S = src + (size_t)yy * sw * cn;
      for (int dx = 0; dx < dw; ++dx)
        for (int c = 0; c < cn; ++c) {
          int v = 0;
          for (int j = 0; j < 4; ++j) {
            const int sxj = imin(imax(xofs[dx] - 1 + j, 0), sw - 1);
            v += S[sxj * cn + c] * ialpha[4 * dx + j];
          }
          rows[k][dx * cn + c] = v;
        }
    }
    uint8_t* D = dst + (size_t)dy * dw * cn;
    for (int i = 0; i < dw * cn; ++i) {
      const int v = rows[0][i] * beta[0] + rows[1][i] * beta[1] + rows[2][i] * beta[2] + rows[3][i] * beta[3];
      D[i] = orc_sat_u8_int((v + (1 << 21)) >> 22);
    }
  }
  for (int k = 0; k < 4; ++k) free(rows[k]);
  free(xofs); free(ialpha);
}

/* cv::interpolateLanczos4 (imgproc/src/resize.cpp) */
static void orc_lanczos4_coeffs(float x, float* coeffs) {
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
  if (x < FLT_EPSILON) {
    for (int i = 0; i < 8; i++) coeffs[i] = 0;
    coeffs[3] = 1;
    return;
  }
  float sum = 0;
  const double y0 = -(x + 3) * 3.1415926535897932384626433832795 * 0.25, s0 = sin(y0), c0 = cos(y0);
  for (int i = 0; i < 8; i++) {
    const double y = -(x + 3 - i) * 3.1415926535897932384626433832795 * 0.25;
    coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
    sum += coeffs[i];
  }
  sum = 1.f / sum;
  for (int i = 0; i < 8; i++) coeffs[i] *= sum;
}

/* One axis of the 8-tap table: offset (first tap = offset - 3) and the 11-bit fixed-point weights
 * saturate_cast<short>(c * 2048).  Also what the HIP launcher builds on the host (same libm). */
ORC_API void orc_lanczos4_axis(int dsize, double scale, int* ofs, short* coef) {
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = cv_floor_f(f);
    f -= s;
    float c[8];
    orc_lanczos4_coeffs(f, c);
    ofs[d] = s;
    for (int k = 0; k < 8; ++k) coef[8 * d + k] = orc_sat_short_round(c[k] * 2048);
  }
}

/* INTER_LANCZOS4, 8-bit: 8 x 8 taps from sx - 3 / sy - 3, columns and rows outside the image
 * replicate the edge (HResizeLanczos4 / clipped row indices), (v + 2^21) >> 22 saturated
 * (VResizeLanczos4 with FixedPtCast<int, uchar, 22>). */
static void orc_resize_lanczos4_u8(const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw,
                                   double scale_x, double scale_y) {
  int* xofs = (int*)malloc(sizeof(int) * dw);
  short* ialpha = (short*)malloc(sizeof(short) * 8 * dw);
  int* yofs = (int*)malloc(sizeof(int) * dh);
  short* ibeta = (short*)malloc(sizeof(short) * 8 * dh);
  orc_lanczos4_axis(dw, scale_x, xofs, ialpha);
  orc_lanczos4_axis(dh, scale_y, yofs, ibeta);
  int* rows[8];
  for (int k = 0; k < 8; ++k) rows[k] = (int*)malloc(sizeof(int) * (size_t)dw * cn);
  for (int dy = 0; dy < dh; ++dy) {
    for (int k = 0; k < 8; ++k) {
      const int yy = imin(imax(yofs[dy] - 3 + k, 0), sh - 1);
      const uint8_t* S = src + (size_t)yy * sw * cn;
      for (int dx = 0; dx < dw; ++dx)
        for (int c = 0; c < cn; ++c) {
          int v = 0;
          for (int j = 0; j < 8; ++j) {
            const int sxj = imin(imax(xofs[dx] - 3 + j, 0), sw - 1);
            v += S[sxj * cn + c] * ialpha[8 * dx + j];
          }
          rows[k][dx * cn + c] = v;
        }
    }
    uint8_t* D = dst + (size_t)dy * dw * cn;
    const short* beta = ibeta + 8 * dy;
    for (int i = 0; i < dw * cn; ++i) {
      int v = 0;
      for (int k = 0; k < 8; ++k) v += rows[k][i] * beta[k];
      D[i] = orc_sat_u8_int((v + (1 << 21)) >> 22);
    }
  }
  for (int k = 0; k < 8; ++k) free(rows[k]);
  free(xofs); free(ialpha); free(yofs); free(ibeta);
}

/* computeResizeAreaTab: the source cells (index, weight) that one axis of an INTER_AREA
 * down-scale accumulates into each destination cell. */
typedef struct { int di, si; float alpha; } orc_area_tab;
static int orc_area_table(int ssize, int dsize, double scale, orc_area_tab* tab) {
  int k = 0;
  for (int dx = 0; dx < dsize; ++dx) {
    const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    const double cellWidth = scale < ssize - fsx1 ? scale : ssize - fsx1;
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = imin(sx2, ssize - 1);
    sx1 = imin(sx1, sx2);
    if (sx1 - fsx1 > 1e-3) { tab[k].di = dx; tab[k].si = sx1 - 1; tab[k++].alpha = (float)((sx1 - fsx1) / cellWidth); }
    for (int sx = sx1; sx < sx2; ++sx) { tab[k].di = dx; tab[k].si = sx; tab[k++].alpha = (float)(1.0 / cellWidth); }
    if (fsx2 - sx2 > 1e-3) {
      double w = fsx2 - sx2;
      if (w > 1.) w = 1.;
      if (w > cellWidth) w = cellWidth;
      tab[k].di = dx; tab[k].si = sx2; tab[k++].alpha = (float)(w / cellWidth);
    }
  }
  return k;
}

/* INTER_AREA, 8-bit.  Down-scaling by integer factors: mean of the iscale_x x iscale_y cell,
 * (sum + 2) >> 2 for 2 x 2 (ResizeAreaFastVec), saturate_cast<uchar>(sum * (1.f/area)) otherwise
 * (resizeAreaFast_Invoker).  Other down-scales: float accumulation of fractionally weighted cells,
 * rows then columns (ResizeArea_Invoker).  Up-scaling (either axis): the bilinear fixed-point path
 * with INTER_AREA's cell-aligned weights. */
static void orc_resize_area_u8(const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw,
                               double scale_x, double scale_y, double inv_scale_x, double inv_scale_y) {
  const int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
  const int fast = fabs(scale_x - iscale_x) < DBL_EPSILON && fabs(scale_y - iscale_y) < DBL_EPSILON;
  if (scale_x >= 1 && scale_y >= 1) {
    if (fast) {
      const int area = iscale_x * iscale_y;
      const float scale = 1.f / area;
      for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x)
          for (int c = 0; c < cn; ++c) {
            int sum = 0;
            for (int sy = 0; sy < iscale_y; ++sy)
              for (int sx = 0; sx < iscale_x; ++sx)
                sum += src[((size_t)(y * iscale_y + sy) * sw + (x * iscale_x + sx)) * cn + c];
            dst[((size_t)y * dw + x) * cn + c] = area == 4 && iscale_x == 2 ? (uint8_t)((sum + 2) >> 2) : orc_sat_u8_float(sum * scale);
          }
      return;
    }
    orc_area_tab* xtab = (orc_area_tab*)malloc(sizeof(orc_area_tab) * ((size_t)sw + 2 * dw + 2));
    orc_area_tab* ytab = (orc_area_tab*)malloc(sizeof(orc_area_tab) * ((size_t)sh + 2 * dh + 2));
    const int xn = orc_area_table(sw, dw, scale_x, xtab), yn = orc_area_table(sh, dh, scale_y, ytab);
    float* buf = (float*)malloc(sizeof(float) * (size_t)dw * cn);
    float* sum = (float*)malloc(sizeof(float) * (size_t)dw * cn);
    int prev_dy = ytab[0].di;
    for (int i = 0; i < dw * cn; ++i) sum[i] = 0;
    for (int j = 0; j < yn; ++j) {
      const int dy = ytab[j].di, sy = ytab[j].si;
      const float beta = ytab[j].alpha;
      const uint8_t* S = src + (size_t)sy * sw * cn;
      for (int i = 0; i < dw * cn; ++i) buf[i] = 0;
      for (int k = 0; k < xn; ++k)
        for (int c = 0; c < cn; ++c) buf[xtab[k].di * cn + c] += S[xtab[k].si * cn + c] * xtab[k].alpha;
      if (dy != prev_dy) {
        uint8_t* D = dst + (size_t)prev_dy * dw * cn;
        for (int i = 0; i < dw * cn; ++i) { D[i] = orc_sat_u8_float(sum[i]); sum[i] = beta * buf[i]; }
        prev_dy = dy;
      } else {
        for (int i = 0; i < dw * cn; ++i) sum[i] += beta * buf[i];
      }
    }
    {
      uint8_t* D = dst + (size_t)prev_dy * dw * cn;
      for (int i = 0; i < dw * cn; ++i) D[i] = orc_sat_u8_float(sum[i]);
    }
    free(xtab); free(ytab); free(buf); free(sum);
    return;
  }
  /* up-scaling on at least one axis: linear arithmetic, area-mode weights on both axes */
  int* xofs = (int*)malloc(sizeof(int) * dw);
  short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
  for (int dx = 0; dx < dw; ++dx) {
    int sx = (int)floor(dx * scale_x);
    float fx = (float)((dx + 1) - (sx + 1) * inv_scale_x);
    fx = fx <= 0 ? 0.f : fx - floorf(fx);
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    ialpha[2 * dx] = orc_sat_short_round((1.f - fx) * 2048);
    ialpha[2 * dx + 1] = orc_sat_short_round(fx * 2048);
  }
  int* rows[2];
  rows[0] = (int*)malloc(sizeof(int) * (size_t)dw * cn);
  rows[1] = (int*)malloc(sizeof(int) * (size_t)dw * cn);
  for (int dy = 0; dy < dh; ++dy) {
    int sy = (int)floor(dy * scale_y);
    float fy = (float)((dy + 1) - (sy + 1) * inv_scale_y);
    fy = fy <= 0 ? 0.f : fy - floorf(fy);
    const short b0 = orc_sat_short_round((1.f - fy) * 2048), b1 = orc_sat_short_round(fy * 2048);
    for (int k = 0; k < 2; ++k) {
      const int yy = imin(imax(sy + k, 0), sh - 1);
      const uint8_t* S = src + (size_t)yy * sw * cn;
      for (int dx = 0; dx < dw; ++dx) {
        const int sx = xofs[dx];
        for (int c = 0; c < cn; ++c) {
          if (sx + 1 < sw) rows[k][dx * cn + c] = S[sx * cn + c] * ialpha[2 * dx] + S[(sx + 1) * cn + c] * ialpha[2 * dx + 1];
          else rows[k][dx * cn + c] = S[sx * cn + c] * 2048;
        }
      }
    }
    uint8_t* D = dst + (size_t)dy * dw * cn;
    for (int i = 0; i < dw * cn; ++i)
      D[i] = (uint8_t)((((b0 * (rows[0][i] >> 4)) >> 16) + ((b1 * (rows[1][i] >> 4)) >> 16) + 2) >> 2);
  }
  free(rows[0]); free(rows[1]); free(xofs); free(ialpha);
}

/* interpolation: cv::InterpolationFlags values 0 = INTER_NEAREST, 1 = INTER_LINEAR, 2 = INTER_CUBIC,
 * 3 = INTER_AREA, 4 = INTER_LANCZOS4 */
ORC_API int orc_resize_u8(const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw, int interpolation) {
  if (interpolation < 0 || interpolation > 4) return 1;
  if (sh == dh && sw == dw) { memcpy(dst, src, (size_t)sh * sw * cn); return 0; }
  const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
  const double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
  if (interpolation == 2) { orc_resize_cubic_u8(src, sh, sw, cn, dst, dh, dw, scale_x, scale_y); return 0; }
  if (interpolation == 4) { orc_resize_lanczos4_u8(src, sh, sw, cn, dst, dh, dw, scale_x, scale_y); return 0; }
  if (interpolation == 3) { orc_resize_area_u8(src, sh, sw, cn, dst, dh, dw, scale_x, scale_y, inv_sx, inv_sy); return 0; }
  if (interpolation == 0) {
    for (int y = 0; y < dh; ++y) {
      int sy = imin((int)floor(y * scale_y), sh - 1);
      for (int x = 0; x < dw; ++x) {
        int sx = imin((int)floor(x * scale_x), sw - 1);
        for (int c = 0; c < cn; ++c) dst[((size_t)y * dw + x) * cn + c] = src[((size_t)sy * sw + sx) * cn + c];
      }
    }
    return 0;
  }
  if (sw == 2 * dw && sh == 2 * dh) {
    for (int y = 0; y < dh; ++y)
      for (int x = 0; x < dw; ++x)
        for (int c = 0; c < cn; ++c) {
          const uint8_t* S0 = src + ((size_t)(2 * y) * sw + 2 * x) * cn + c;
          const uint8_t* S1 = S0 + (size_t)sw * cn;
          dst[((size_t)y * dw + x) * cn + c] = (uint8_t)((S0[0] + S0[cn] + S1[0] + S1[cn] + 2) >> 2);
        }
    return 0;
  }
  int* xofs = (int*)malloc(sizeof(int) * dw);
  short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    ialpha[2 * dx] = orc_sat_short_round((1.f - fx) * 2048);
    ialpha[2 * dx + 1] = orc_sat_short_round(fx * 2048);
  }
  int* rows[2];
  rows[0] = (int*)malloc(sizeof(int) * (size_t)dw * cn);
  rows[1] = (int*)malloc(sizeof(int) * (size_t)dw * cn);
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor_f(fy);
    fy -= sy;
    const short b0 = orc_sat_short_round((1.f - fy) * 2048), b1 = orc_sat_short_round(fy * 2048);
    for (int k = 0; k < 2; ++k) {
      int yy = imin(imax(sy + k, 0), sh - 1);
      const uint8_t* S = src + (size_t)yy * sw * cn;
      for (int dx = 0; dx < dw; ++dx) {
        int sx = xofs[dx];
        for (int c = 0; c < cn; ++c) {
          if (sx + 1 < sw) rows[k][dx * cn + c] = S[sx * cn + c] * ialpha[2 * dx] + S[(sx + 1) * cn + c] * ialpha[2 * dx + 1];
          else rows[k][dx * cn + c] = S[sx * cn + c] * 2048;
        }
      }
    }
    uint8_t* D = dst + (size_t)dy * dw * cn;
    for (int i = 0; i < dw * cn; ++i)
      D[i] = (uint8_t)((((b0 * (rows[0][i] >> 4)) >> 16) + ((b1 * (rows[1][i] >> 4)) >> 16) + 2) >> 2);
  }
  free(rows[0]); free(rows[1]); free(xofs); free(ialpha);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * ConvertColor op -- ConvertColorKernel::execute,
 * /root/reference/scannertools/scannertools_cpp/imgproc/convert_color_kernel.cpp:252-285:
 * cv::cvtColor(img, out, code) on U8 frames, code chosen by name from the table at :10-209.
 * Restated codes (cv::ColorConversionCodes values): BGR2RGB/RGB2BGR (4), BGR2GRAY (6),
 * RGB2GRAY (7), GRAY2BGR/GRAY2RGB (8), BGR2YCrCb (36), RGB2YCrCb (37), YCrCb2BGR (38),
 * YCrCb2RGB (39) [14-bit RGB2YCrCb_i / YCrCb2RGB_i tables], BGR2HSV (40).  8-bit arithmetic of OpenCV
 * (imgproc/src/color*.cpp): gray = (c_b*B + c_g*G + c_r*R + half) >> bits with the 14/15-bit
 * tables of orc_gray_u8; RGB2HSV_b with hsv_shift = 12: v = max, s = (diff*sdiv[v] + 2^11) >> 12,
 * h = sector offset + difference, scaled by hdiv180[diff], +180 if negative, where
 * sdiv[i] = cvRound((255 << 12)/(1.*i)), hdiv180[i] = cvRound((180 << 12)/(6.*i)), both 0 at i = 0.
 * PARITY UNPINNED against real OpenCV output (integer arithmetic; pinned by primaries, grays and
 * hue-sector identities in tests/).
 * ------------------------------------------------------------------------------------------ */
enum { ORC_BGR2RGB = 4, ORC_BGR2GRAY = 6, ORC_RGB2GRAY = 7, ORC_GRAY2BGR = 8, ORC_BGR2YCrCb = 36, ORC_RGB2YCrCb = 37,
       ORC_YCrCb2BGR = 38, ORC_YCrCb2RGB = 39, ORC_BGR2HSV = 40, ORC_RGB2HSV = 41, ORC_HSV2BGR = 54, ORC_HSV2RGB = 55,
       ORC_BGR2HSV_FULL = 66, ORC_RGB2HSV_FULL = 67, ORC_HSV2BGR_FULL = 70, ORC_HSV2RGB_FULL = 71,
       ORC_BGR2YUV = 82, ORC_RGB2YUV = 83, ORC_YUV2BGR = 84, ORC_YUV2RGB = 85,
       ORC_BGR2HLS = 52, ORC_RGB2HLS = 53, ORC_HLS2BGR = 60, ORC_HLS2RGB = 61,
       ORC_BGR2HLS_FULL = 68, ORC_RGB2HLS_FULL = 69, ORC_HLS2BGR_FULL = 72, ORC_HLS2RGB_FULL = 73 };

/* Further codes of the reference's table (convert_color_kernel.cpp:60-93), restated from
 * imgproc/src/color_hsv.simd.hpp and color_yuv.simd.hpp:
 *   RGB2HSV (41), BGR/RGB2HSV_FULL (66, 67): RGB2HSV_b with blue index 2 / hue range 256
 *     (hdiv256[i] = cvRound((256 << 12)/(6.*i)));
 *   HSV2BGR/RGB (54, 55), _FULL (70, 71): HSV2RGB_b = bytes -> float (h, s/255, v/255) ->
 *     HSV2RGB_native (h *= 6/hrange; wrapped into [0, 6); sector = floor(h); tab = {v, v(1-s),
 *     v(1-s h), v(1-s(1-h))}; sector table {{1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}} giving
 *     (b, g, r)) -> saturate_cast<uchar>(x * 255), float arithmetic throughout;
 *   BGR/RGB2YUV (82, 83): RGB2YCrCb_i with {R2YI, G2YI, B2YI, R2VI, B2UI} = {4899, 9617, 1868, 14369,
 *     8061} and the chroma pair stored as (U, V) = (Cb-like, Cr-like);
 *   YUV2BGR/RGB (84, 85): YCrCb2RGB_i with {V2RI, V2GI, U2GI, U2BI} = {18678, -9519, -6472, 33292}.
 * PARITY UNPINNED against real OpenCV output like the codes above. */

ORC_API int orc_cvt_out_channels(int code, int in_channels) {
  switch (code) {
    case ORC_BGR2RGB: return in_channels == 3 ? 3 : -1;
    case ORC_BGR2GRAY: case ORC_RGB2GRAY: return in_channels == 3 ? 1 : -1;
    case ORC_GRAY2BGR: return in_channels == 1 ? 3 : -1;
    case ORC_BGR2HSV: case ORC_BGR2YCrCb: case ORC_RGB2YCrCb: case ORC_YCrCb2BGR: case ORC_YCrCb2RGB:
    case ORC_RGB2HSV: case ORC_HSV2BGR: case ORC_HSV2RGB: case ORC_BGR2HSV_FULL: case ORC_RGB2HSV_FULL:
    case ORC_HSV2BGR_FULL: case ORC_HSV2RGB_FULL: case ORC_BGR2YUV: case ORC_RGB2YUV: case ORC_YUV2BGR: case ORC_YUV2RGB:
    case ORC_BGR2HLS: case ORC_RGB2HLS: case ORC_HLS2BGR: case ORC_HLS2RGB:
    case ORC_BGR2HLS_FULL: case ORC_RGB2HLS_FULL: case ORC_HLS2BGR_FULL: case ORC_HLS2RGB_FULL:
      return in_channels == 3 ? 3 : -1;
    case 32: case 33: case 34: case 35: return in_channels == 3 ? 3 : -1;   /* BGR2XYZ, RGB2XYZ, XYZ2BGR, XYZ2RGB */
    /* channel layout family (cv::cvtColor codes 0..3, 5, 9..31) */
    case 0: case 2: return in_channels == 3 ? 4 : -1;          /* BGR2BGRA, BGR2RGBA */
    case 1: case 3: return in_channels == 4 ? 3 : -1;          /* BGRA2BGR, RGBA2BGR */
    case 5: return in_channels == 4 ? 4 : -1;                  /* BGRA2RGBA */
    case 9: return in_channels == 1 ? 4 : -1;                  /* GRAY2BGRA */
    case 10: case 11: return in_channels == 4 ? 1 : -1;        /* BGRA2GRAY, RGBA2GRAY */
    case 12: case 13: case 22: case 23: return in_channels == 3 ? 2 : -1;   /* BGR / RGB -> BGR565 / BGR555 */
    case 14: case 15: case 24: case 25: return in_channels == 2 ? 3 : -1;   /* BGR565 / BGR555 -> BGR / RGB */
    case 16: case 17: case 26: case 27: return in_channels == 4 ? 2 : -1;   /* BGRA / RGBA -> packed */
    case 18: case 19: case 28: case 29: return in_channels == 2 ? 4 : -1;   /* packed -> BGRA / RGBA */
    case 20: case 30: return in_channels == 1 ? 2 : -1;                     /* GRAY -> packed */
    case 21: case 31: return in_channels == 2 ? 1 : -1;                     /* packed -> GRAY */
    default: return -1;
  }
}

/* The functors of OpenCV's color_rgb (restated per functor, with their (srccn, dstcn, blueIdx, greenBits) parameters):
 * RGB2RGB<uchar>: copies / swaps the colour channels, alpha = src alpha or 255.
 * RGB2RGB5x5: ushort = (b >> 3) | ((g & ~3) << 3) | ((r & ~7) << 8) for 6 green bits,
 *             (b >> 3) | ((g & ~7) << 2) | ((r & ~7) << 7) | (srccn == 4 && alpha ? 0x8000 : 0) for 5.
 * RGB5x52RGB: b = t << 3, g = (t >> 3) & ~3, r = (t >> 8) & ~7 (alpha 255) for 6 green bits,
 *             b = t << 3, g = (t >> 2) & ~7, r = (t >> 7) & ~7 (alpha = t & 0x8000 ? 255 : 0) for 5, each as uchar.
 * Gray2RGB5x5: t |-> the same packing with b = g = r = t.   RGB5x52Gray: CV_DESCALE(b * 1868 + g * 9617 + r * 4899, 14). */
static void orc_rgb2rgb(const uint8_t* src, size_t n, int scn, int dcn, int bidx, uint8_t* dst) {
  for (size_t i = 0; i < n; ++i, src += scn, dst += dcn) {
    const uint8_t t0 = src[bidx], t1 = src[1], t2 = src[bidx ^ 2];
    dst[0] = t0; dst[1] = t1; dst[2] = t2;
    if (dcn == 4) dst[3] = scn == 4 ? src[3] : 255;
  }
}
static void orc_rgb2rgb5x5(const uint8_t* src, size_t n, int scn, int bidx, int green_bits, uint8_t* dst) {
  for (size_t i = 0; i < n; ++i, src += scn) {
    unsigned t;
    if (green_bits == 6) t = (unsigned)(src[bidx] >> 3) | ((unsigned)(src[1] & ~3) << 3) | ((unsigned)(src[bidx ^ 2] & ~7) << 8);
    else t = (unsigned)(src[bidx] >> 3) | ((unsigned)(src[1] & ~7) << 2) | ((unsigned)(src[bidx ^ 2] & ~7) << 7) | (scn == 4 && src[3] ? 0x8000u : 0u);
    dst[2 * i] = (uint8_t)t; dst[2 * i + 1] = (uint8_t)(t >> 8);
  }
}
static void orc_rgb5x52rgb(const uint8_t* src, size_t n, int dcn, int bidx, int green_bits, uint8_t* dst) {
  for (size_t i = 0; i < n; ++i, dst += dcn) {
    const unsigned t = src[2 * i] | ((unsigned)src[2 * i + 1] << 8);
    if (green_bits == 6) {
      dst[bidx] = (uint8_t)(t << 3); dst[1] = (uint8_t)((t >> 3) & ~3u); dst[bidx ^ 2] = (uint8_t)((t >> 8) & ~7u);
      if (dcn == 4) dst[3] = 255;
    } else {
      dst[bidx] = (uint8_t)(t << 3); dst[1] = (uint8_t)((t >> 2) & ~7u); dst[bidx ^ 2] = (uint8_t)((t >> 7) & ~7u);
      if (dcn == 4) dst[3] = (t & 0x8000u) ? 255 : 0;
    }
  }
}

/* HSV2RGB_native (color_hsv.simd.hpp); hscale = 6.f / hrange */
static void orc_hsv2rgb_native(float h, float s, float v, float hscale, float* b, float* g, float* r) {
  if (s == 0) { *b = *g = *r = v; return; }
  static const int sector_data[][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
  float tab[4];
  h *= hscale;
  h = fmodf(h, 6.f);
  int sector = (int)floorf(h);
  h -= sector;
  if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
  tab[0] = v;
  tab[1] = v * (1.f - s);
  tab[2] = v * (1.f - s * h);
  tab[3] = v * (1.f - s * (1.f - h));
  *b = tab[sector_data[sector][0]];
  *g = tab[sector_data[sector][1]];
  *r = tab[sector_data[sector][2]];
}

/* YUV 4:2:0 / 4:2:2 sources -> RGB (OpenCV color_yuv: YUV420sp2RGB888Invoker, YUV420p2RGB888Invoker, YUV422toRGB888Invoker and
 * their 8888 twins).  Walks the frame the way those invokers do -- two rows and two columns at a time, one (u, v) pair
 * per block -- with the BT.601 constants derived from their decimal values (1.164, 2.018, 0.391, 0.813, 1.596) x 2^20. */
static int orc_yuv_layout(int code, int* kind, int* bidx, int* uidx, int* yidx, int* dcn) {
  if (code >= 90 && code <= 97) { const int c = code - 90; *kind = 0; *bidx = (c & 1) ? 0 : 2; *uidx = (c >> 1) & 1; *yidx = 0; *dcn = c >= 4 ? 4 : 3; return 1; }
  if (code >= 98 && code <= 105) { const int c = code - 98; *kind = 1; *bidx = (c & 1) ? 0 : 2; *uidx = ((c >> 1) & 1) ? 0 : 1; *yidx = 0; *dcn = c >= 4 ? 4 : 3; return 1; }
  if (code == 106) { *kind = 3; *bidx = *uidx = *yidx = 0; *dcn = 1; return 1; }
  static const int t[][5] = {{107, 2, 0, 1, 3}, {108, 0, 0, 1, 3}, {111, 2, 0, 1, 4}, {112, 0, 0, 1, 4}, {115, 2, 1, 0, 3}, {116, 0, 1, 0, 3},
                             {117, 2, 3, 0, 3}, {118, 0, 3, 0, 3}, {119, 2, 1, 0, 4}, {120, 0, 1, 0, 4}, {121, 2, 3, 0, 4}, {122, 0, 3, 0, 4}};
  for (unsigned i = 0; i < sizeof(t) / sizeof(t[0]); ++i)
    if (t[i][0] == code) { *kind = 2; *bidx = t[i][1]; *uidx = t[i][2]; *yidx = t[i][3]; *dcn = t[i][4]; return 1; }
  if (code == 123) { *kind = 4; *bidx = 0; *uidx = 0; *yidx = 1; *dcn = 1; return 1; }
  if (code == 124) { *kind = 4; *bidx = 0; *uidx = 1; *yidx = 0; *dcn = 1; return 1; }
  return 0;
}

ORC_API int orc_cvt_color_out_shape(int code, int in_h, int in_w, int in_channels, int* oh, int* ow, int* oc) {
  int kind, bidx, uidx, yidx, dcn;
  if (orc_yuv_layout(code, &kind, &bidx, &uidx, &yidx, &dcn)) {
    if (kind == 2 || kind == 4) {
      if (in_channels != 2 || in_w % 2 || in_h <= 0 || in_w <= 0) return -1;
      *oh = in_h;
    } else {
      if (in_channels != 1 || in_w % 2 || in_h % 3 || in_h <= 0 || in_w <= 0) return -1;
      *oh = in_h * 2 / 3;
    }
    *ow = in_w; *oc = dcn;
    return 0;
  }
  const int c = orc_cvt_out_channels(code, in_channels);
  if (c < 0) return -1;
  *oh = in_h; *ow = in_w; *oc = c;
  return 0;
}

static void orc_yuv_px(int Y, int u, int v, int bidx, int dcn, uint8_t* o) {
  const int CY = (int)(1.164 * 1048576), CUB = (int)(2.018 * 1048576), CUG = -(int)(0.391 * 1048576), CVG = -(int)(0.813 * 1048576),
            CVR = (int)(1.596 * 1048576);
  const int y = (Y - 16 > 0 ? Y - 16 : 0) * CY;
  const int ruv = (1 << 19) + CVR * v, guv = (1 << 19) + CVG * v + CUG * u, buv = (1 << 19) + CUB * u;
  int r = (y + ruv) >> 20, g = (y + guv) >> 20, b = (y + buv) >> 20;
  o[bidx] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
  o[1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
  o[bidx ^ 2] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
  if (dcn == 4) o[3] = 255;
}

static int orc_cvt_yuv(const uint8_t* src, int h, int w, int cn, int code, uint8_t* dst) {
  int kind, bidx, uidx, yidx, dcn, H, W, C;
  if (!orc_yuv_layout(code, &kind, &bidx, &uidx, &yidx, &dcn) || orc_cvt_color_out_shape(code, h, w, cn, &H, &W, &C)) return 1;
  if (kind == 3) { memcpy(dst, src, (size_t)H * W); return 0; }
  if (kind == 4) { for (size_t i = 0; i < (size_t)H * W; ++i) dst[i] = src[2 * (i & ~(size_t)1) + yidx + 2 * (i & 1)]; return 0; }
  if (kind == 2) {
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; x += 2) {
        const uint8_t* g = src + ((size_t)y * W + x) * 2;
        const int u = g[uidx] - 128, v = g[uidx ^ 2] - 128;
        orc_yuv_px(g[yidx], u, v, bidx, dcn, dst + ((size_t)y * W + x) * dcn);
        orc_yuv_px(g[yidx + 2], u, v, bidx, dcn, dst + ((size_t)y * W + x + 1) * dcn);
      }
    return 0;
  }
  const uint8_t* chroma = src + (size_t)H * W;
  for (int y = 0; y < H; y += 2)
    for (int x = 0; x < W; x += 2) {
      int u, v;
      if (kind == 0) {
        const uint8_t* uv = chroma + (size_t)(y / 2) * W + x;
        u = uv[uidx] - 128; v = uv[1 - uidx] - 128;
      } else {
        const size_t q = (size_t)(H / 2) * (W / 2), o = (size_t)(y / 2) * (W / 2) + x / 2;
        u = (uidx == 0 ? chroma[o] : chroma[q + o]) - 128;
        v = (uidx == 0 ? chroma[q + o] : chroma[o]) - 128;
      }
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx)
          orc_yuv_px(src[(size_t)(y + dy) * W + x + dx], u, v, bidx, dcn, dst + ((size_t)(y + dy) * W + x + dx) * dcn);
    }
  return 0;
}

ORC_API int orc_cvt_color_u8(const uint8_t* src, int h, int w, int cn, int code, int gray_bits, uint8_t* dst) {
  if (code >= 90 && code <= 124) return orc_cvt_yuv(src, h, w, cn, code, dst);
  if (orc_cvt_out_channels(code, cn) < 0) return 1;
  const size_t n = (size_t)h * w;
  if (code <= 3 || code == 5) {        /* RGB2RGB<uchar>(scn, dcn, blueIdx): blueIdx 2 swaps */
    orc_rgb2rgb(src, n, cn, (code == 0 || code == 2) ? 4 : (code == 5 ? 4 : 3), (code == 2 || code == 3 || code == 5) ? 2 : 0, dst);
  } else if (code == 9) {              /* Gray2RGB<uchar>(4) */
    for (size_t i = 0; i < n; ++i) { dst[4 * i] = dst[4 * i + 1] = dst[4 * i + 2] = src[i]; dst[4 * i + 3] = 255; }
  } else if (code == 10 || code == 11) {  /* RGB2Gray<uchar>(4, blueIdx) */
    int cb, cg, cr, bits = gray_bits;
    if (bits == 14) { cb = 1868; cg = 9617; cr = 4899; } else { bits = 15; cb = 3735; cg = 19235; cr = 9798; }
    const int bi = code == 10 ? 0 : 2;
    for (size_t i = 0; i < n; ++i)
      dst[i] = (uint8_t)((src[4 * i + bi] * cb + src[4 * i + 1] * cg + src[4 * i + (bi ^ 2)] * cr + (1 << (bits - 1))) >> bits);
  } else if (code >= 12 && code <= 31) {
    const int gb = code >= 22 ? 5 : 6, c = code >= 22 ? code - 22 : code - 12;
    if (c == 0 || c == 1 || c == 4 || c == 5) orc_rgb2rgb5x5(src, n, cn, (c & 1) ? 2 : 0, gb, dst);
    else if (c == 2 || c == 3 || c == 6 || c == 7) orc_rgb5x52rgb(src, n, c >= 6 ? 4 : 3, (c & 1) ? 2 : 0, gb, dst);
    else if (c == 8) {                 /* Gray2RGB5x5 */
      for (size_t i = 0; i < n; ++i) {
        unsigned t = src[i];
        if (gb == 6) t = (t >> 3) | ((t & ~3u) << 3) | ((t & ~7u) << 8);
        else { t >>= 3; t = t | (t << 5) | (t << 10); }
        dst[2 * i] = (uint8_t)t; dst[2 * i + 1] = (uint8_t)(t >> 8);
      }
    } else {                           /* RGB5x52Gray */
      for (size_t i = 0; i < n; ++i) {
        const unsigned t = src[2 * i] | ((unsigned)src[2 * i + 1] << 8);
        const int v = gb == 6 ? (int)((t << 3) & 0xf8) * 1868 + (int)((t >> 3) & 0xfc) * 9617 + (int)((t >> 8) & 0xf8) * 4899
                              : (int)((t << 3) & 0xf8) * 1868 + (int)((t >> 2) & 0xf8) * 9617 + (int)((t >> 7) & 0xf8) * 4899;
        dst[i] = (uint8_t)((v + (1 << 13)) >> 14);
      }
    }
  } else if (code >= 32 && code <= 35) {
    /* RGB2XYZ_i<uchar> / XYZ2RGB_i<uchar> (color_lab.cpp): the D65 matrices sRGB2XYZ_D65 / XYZ2sRGB_D65 scaled by 1 << 12
       and rounded -- recomputed here from the float matrices rather than typed in --, with the blue / red columns (rows)
       exchanged for blueIdx 0; CV_DESCALE(.., 12), saturate_cast<uchar>. */
    static const double fwd[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
    static const double inv[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};
    int C[9];
    const int to_xyz = code <= 33, bidx = (code == 32 || code == 34) ? 0 : 2;
    for (int k = 0; k < 9; ++k) C[k] = (int)lrint((to_xyz ? fwd[k] : inv[k]) * 4096.);
    if (bidx == 0) {
      if (to_xyz) { int t; t = C[0]; C[0] = C[2]; C[2] = t; t = C[3]; C[3] = C[5]; C[5] = t; t = C[6]; C[6] = C[8]; C[8] = t; }
      else { int t; for (int k = 0; k < 3; ++k) { t = C[k]; C[k] = C[6 + k]; C[6 + k] = t; } }
    }
    for (size_t i = 0; i < n; ++i) {
      const uint8_t* p = src + 3 * i;
      for (int k = 0; k < 3; ++k) {
        const int v = (p[0] * C[3 * k] + p[1] * C[3 * k + 1] + p[2] * C[3 * k + 2] + (1 << 11)) >> 12;
        dst[3 * i + k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
      }
    }
  } else if (code == ORC_BGR2RGB) {
    for (size_t i = 0; i < n; ++i) { dst[3 * i] = src[3 * i + 2]; dst[3 * i + 1] = src[3 * i + 1]; dst[3 * i + 2] = src[3 * i]; }
  } else if (code == ORC_BGR2GRAY || code == ORC_RGB2GRAY) {
    int cb, cg, cr, bits = gray_bits;
    if (bits == 14) { cb = 1868; cg = 9617; cr = 4899; } else { bits = 15; cb = 3735; cg = 19235; cr = 9798; }
    const int rnd = 1 << (bits - 1);
    const int bi = code == ORC_BGR2GRAY ? 0 : 2;  /* byte holding blue */
    for (size_t i = 0; i < n; ++i)
      dst[i] = (uint8_t)((src[3 * i + bi] * cb + src[3 * i + 1] * cg + src[3 * i + (bi ^ 2)] * cr + rnd) >> bits);
  } else if (code == ORC_GRAY2BGR) {
    for (size_t i = 0; i < n; ++i) dst[3 * i] = dst[3 * i + 1] = dst[3 * i + 2] = src[i];
  } else if (code == ORC_BGR2YCrCb || code == ORC_RGB2YCrCb) {
    /* RGB2YCrCb_i<uchar>: yuv_shift = 14, {R2YI, G2YI, B2YI, YCRI, YCBI} = {4899, 9617, 1868, 11682, 9241} */
    const int bidx = code == ORC_BGR2YCrCb ? 0 : 2, sh = 14, delta = 128 * (1 << 14);
    const int C0 = bidx == 0 ? 1868 : 4899, C1 = 9617, C2 = bidx == 0 ? 4899 : 1868, C3 = 11682, C4 = 9241;
    for (size_t i = 0; i < n; ++i) {
      const uint8_t* p = src + 3 * i;
      const int Y = (p[0] * C0 + p[1] * C1 + p[2] * C2 + (1 << (sh - 1))) >> sh;
      const int Cr = ((p[bidx ^ 2] - Y) * C3 + delta + (1 << (sh - 1))) >> sh;
      const int Cb = ((p[bidx] - Y) * C4 + delta + (1 << (sh - 1))) >> sh;
      dst[3 * i] = (uint8_t)(Y < 0 ? 0 : (Y > 255 ? 255 : Y));
      dst[3 * i + 1] = (uint8_t)(Cr < 0 ? 0 : (Cr > 255 ? 255 : Cr));
      dst[3 * i + 2] = (uint8_t)(Cb < 0 ? 0 : (Cb > 255 ? 255 : Cb));
    }
  } else if (code == ORC_YCrCb2BGR || code == ORC_YCrCb2RGB) {
    /* YCrCb2RGB_i<uchar>: {CR2RI, CR2GI, CB2GI, CB2BI} = {22987, -11698, -5636, 29049} */
    const int bidx = code == ORC_YCrCb2BGR ? 0 : 2, sh = 14, rnd = 1 << (sh - 1);
    for (size_t i = 0; i < n; ++i) {
      const int Y = src[3 * i], Cr = src[3 * i + 1] - 128, Cb = src[3 * i + 2] - 128;
      const int b = Y + ((Cb * 29049 + rnd) >> sh);
      const int g = Y + ((Cb * -5636 + Cr * -11698 + rnd) >> sh);
      const int r = Y + ((Cr * 22987 + rnd) >> sh);
      dst[3 * i + bidx] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
      dst[3 * i + 1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
      dst[3 * i + (bidx ^ 2)] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
  } else if (code == ORC_BGR2YUV || code == ORC_RGB2YUV) {
    const int bidx = code == ORC_BGR2YUV ? 0 : 2, sh = 14, delta = 128 * (1 << 14);
    const int C0 = bidx == 0 ? 1868 : 4899, C1 = 9617, C2 = bidx == 0 ? 4899 : 1868, C3 = 14369, C4 = 8061;
    for (size_t i = 0; i < n; ++i) {
      const uint8_t* p = src + 3 * i;
      const int Y = (p[0] * C0 + p[1] * C1 + p[2] * C2 + (1 << (sh - 1))) >> sh;
      const int V = ((p[bidx ^ 2] - Y) * C3 + delta + (1 << (sh - 1))) >> sh;
      const int U = ((p[bidx] - Y) * C4 + delta + (1 << (sh - 1))) >> sh;
      dst[3 * i] = orc_sat_u8_int(Y);
      dst[3 * i + 1] = orc_sat_u8_int(U);   /* yuvOrder: dst[i + 2 - 1] = Cb */
      dst[3 * i + 2] = orc_sat_u8_int(V);   /*           dst[i + 1 + 1] = Cr */
    }
  } else if (code == ORC_YUV2BGR || code == ORC_YUV2RGB) {
    const int bidx = code == ORC_YUV2BGR ? 0 : 2, sh = 14, rnd = 1 << (sh - 1);
    for (size_t i = 0; i < n; ++i) {
      const int Y = src[3 * i], U = src[3 * i + 1] - 128, V = src[3 * i + 2] - 128;
      const int b = Y + ((U * 33292 + rnd) >> sh);
      const int g = Y + ((U * -6472 + V * -9519 + rnd) >> sh);
      const int r = Y + ((V * 18678 + rnd) >> sh);
      dst[3 * i + bidx] = orc_sat_u8_int(b);
      dst[3 * i + 1] = orc_sat_u8_int(g);
      dst[3 * i + (bidx ^ 2)] = orc_sat_u8_int(r);
    }
  } else if (code == ORC_HSV2BGR || code == ORC_HSV2RGB || code == ORC_HSV2BGR_FULL || code == ORC_HSV2RGB_FULL) {
    const int bidx = (code == ORC_HSV2BGR || code == ORC_HSV2BGR_FULL) ? 0 : 2;
    const int hrange = (code == ORC_HSV2BGR || code == ORC_HSV2RGB) ? 180 : 255;  /* cvtColor passes 255 for _FULL on 8U */
    const float hscale = 6.f / hrange;
    for (size_t i = 0; i < n; ++i) {
      const float hh = src[3 * i], ss = src[3 * i + 1] * (1.f / 255.f), vv = src[3 * i + 2] * (1.f / 255.f);
      float b, g, r;
      orc_hsv2rgb_native(hh, ss, vv, hscale, &b, &g, &r);
      dst[3 * i + bidx] = orc_sat_u8_float(b * 255.f);
      dst[3 * i + 1] = orc_sat_u8_float(g * 255.f);
      dst[3 * i + (bidx ^ 2)] = orc_sat_u8_float(r * 255.f);
    }
  } else if (code == ORC_BGR2HLS || code == ORC_RGB2HLS || code == ORC_BGR2HLS_FULL || code == ORC_RGB2HLS_FULL) {
    /* RGB2HLS_b (imgproc/src/color_hsv.simd.hpp): bytes x (1/255) -> RGB2HLS_f, scalar formulation -> H =
     * saturate_cast<uchar>(h x hrange/360) with hrange 180 (256 for _FULL), L and S x 255.  The vector bodies of OpenCV 4
     * compute the same quantities with a fused multiply-add and 2 - (max + min); where a build takes them the last bit of
     * a float can differ.  PARITY UNPINNED. */
    const int bidx = (code == ORC_BGR2HLS || code == ORC_BGR2HLS_FULL) ? 0 : 2;
    const float hscale = ((code == ORC_BGR2HLS || code == ORC_RGB2HLS) ? 180.f : 256.f) / 360.f;
    for (size_t i = 0; i < n; ++i) {
      const float b = src[3 * i + bidx] * (1.f / 255.f), g = src[3 * i + 1] * (1.f / 255.f), r = src[3 * i + (bidx ^ 2)] * (1.f / 255.f);
      float h = 0.f, s = 0.f, l, vmin, vmax, diff;
      vmax = vmin = r;
      if (vmax < g) vmax = g;
      if (vmax < b) vmax = b;
      if (vmin > g) vmin = g;
      if (vmin > b) vmin = b;
      diff = vmax - vmin;
      l = (vmax + vmin) * 0.5f;
      if (diff > FLT_EPSILON) {
        s = l < 0.5f ? diff / (vmax + vmin) : diff / (2 - vmax - vmin);
        diff = 60.f / diff;
        if (vmax == r) h = (g - b) * diff;
        else if (vmax == g) h = (b - r) * diff + 120.f;
        else h = (r - g) * diff + 240.f;
        if (h < 0.f) h += 360.f;
      }
      dst[3 * i] = orc_sat_u8_float(h * hscale);
      dst[3 * i + 1] = orc_sat_u8_float(l * 255.f);
      dst[3 * i + 2] = orc_sat_u8_float(s * 255.f);
    }
  } else if (code == ORC_HLS2BGR || code == ORC_HLS2RGB || code == ORC_HLS2BGR_FULL || code == ORC_HLS2RGB_FULL) {
    /* HLS2RGB_b: bytes -> (h, l/255, s/255) -> HLS2RGB_native (p2 = l <= 0.5 ? l (1 + s) : l + s - l s; p1 = 2 l - p2; h x
     * 6/hrange wrapped into [0, 6); tab = {p2, p1, p1 + (p2 - p1)(1 - h), p1 + (p2 - p1) h}; the sector table of HSV2RGB) ->
     * saturate_cast<uchar>(x * 255); hrange 180 (255 for _FULL, as cvtColor passes it for 8-bit data).  PARITY UNPINNED. */
    static const int sector_data[][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    const int bidx = (code == ORC_HLS2BGR || code == ORC_HLS2BGR_FULL) ? 0 : 2;
    const float hscale = 6.f / ((code == ORC_HLS2BGR || code == ORC_HLS2RGB) ? 180 : 255);
    for (size_t i = 0; i < n; ++i) {
      float h = src[3 * i];
      const float l = src[3 * i + 1] * (1.f / 255.f), s = src[3 * i + 2] * (1.f / 255.f);
      float b, g, r;
      if (s == 0) {
        b = g = r = l;
      } else {
        float tab[4];
        const float p2 = l <= 0.5f ? l * (1 + s) : l + s - l * s;
        const float p1 = 2 * l - p2;
        h *= hscale;
        while (h < 0) h += 6;
        while (h >= 6) h -= 6;
        int sector = (int)floorf(h);
        h -= sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        tab[0] = p2;
        tab[1] = p1;
        tab[2] = p1 + (p2 - p1) * (1 - h);
        tab[3] = p1 + (p2 - p1) * h;
        b = tab[sector_data[sector][0]];
        g = tab[sector_data[sector][1]];
        r = tab[sector_data[sector][2]];
      }
      dst[3 * i + bidx] = orc_sat_u8_float(b * 255.f);
      dst[3 * i + 1] = orc_sat_u8_float(g * 255.f);
      dst[3 * i + (bidx ^ 2)] = orc_sat_u8_float(r * 255.f);
    }
  } else {  /* BGR2HSV / RGB2HSV, hrange 180 (256 for _FULL) */
    const int hsv_shift = 12;
    const int bidx = (code == ORC_BGR2HSV || code == ORC_BGR2HSV_FULL) ? 0 : 2;
    const int hr = (code == ORC_BGR2HSV || code == ORC_RGB2HSV) ? 180 : 256;
    int sdiv[256], hdiv[256];
    sdiv[0] = hdiv[0] = 0;
    for (int i = 1; i < 256; ++i) {
      sdiv[i] = (int)lrint((255 << hsv_shift) / (1. * i));
      hdiv[i] = (int)lrint((hr << hsv_shift) / (6. * i));
    }
    for (size_t i = 0; i < n; ++i) {
      const int b = src[3 * i + bidx], g = src[3 * i + 1], r = src[3 * i + (bidx ^ 2)];
      int v = b, vmin = b;
      if (g > v) v = g;
      if (r > v) v = r;
      if (g < vmin) vmin = g;
      if (r < vmin) vmin = r;
      const int diff = v - vmin;
      const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
      const int s = (diff * sdiv[v] + (1 << (hsv_shift - 1))) >> hsv_shift;
      int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
      hh = (hh * hdiv[diff] + (1 << (hsv_shift - 1))) >> hsv_shift;
      hh += hh < 0 ? hr : 0;
      dst[3 * i] = (uint8_t)(hh < 0 ? 0 : (hh > 255 ? 255 : hh));
      dst[3 * i + 1] = (uint8_t)s;
      dst[3 * i + 2] = (uint8_t)v;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * CPM2Input op -- CPM2InputKernel::execute,
 * /root/reference/scannertools_caffe/scannertools_caffe_cpp/cpm2_input_kernel_gpu.cpp:104-140, stated
 * with the OpenCV CPU functions of the same names, one after the other as the reference calls them:
 * cvtColor(RGB2BGR) -> resize(INTER_CUBIC) to ((int)(w*scale), (int)(h*scale)) -> copyMakeBorder(bottom,
 * right up to a multiple of 8, BORDER_CONSTANT 128) -> convertTo(CV_32FC3, 1/256, -0.5) -> split into a
 * planar (3, net_h, net_w) array.  (The reference file uses cv::cuda::resize, whose bicubic is another
 * filter; not the parity target -- SURVEY.md section 0.4.)  PARITY UNPINNED against real OpenCV output like
 * the Resize op whose restatement it reuses.
 * ------------------------------------------------------------------------------------------ */
ORC_API int orc_cpm2_geometry(int h, int w, float scale, int* rh, int* rw, int* nh, int* nw) {
  if (h <= 0 || w <= 0 || !(scale > 0)) return 1;
  const int resize_width = (int)(w * scale), resize_height = (int)(h * scale);  /* :48-49 */
  if (resize_width <= 0 || resize_height <= 0) return 1;
  const int width_padding = (resize_width % 8) ? 8 - (resize_width % 8) : 0;
  const int height_padding = (resize_height % 8) ? 8 - (resize_height % 8) : 0;
  *rh = resize_height; *rw = resize_width;
  *nh = resize_height + height_padding; *nw = resize_width + width_padding;
  return 0;
}

ORC_API int orc_cpm2_input(const uint8_t* rgb, int h, int w, float scale, float* out) {
  int rh, rw, nh, nw;
  if (orc_cpm2_geometry(h, w, scale, &rh, &rw, &nh, &nw)) return 1;
  uint8_t* bgr = (uint8_t*)malloc((size_t)h * w * 3);
  uint8_t* resized = (uint8_t*)malloc((size_t)rh * rw * 3);
  orc_cvt_color_u8(rgb, h, w, 3, ORC_BGR2RGB, 15, bgr);
  orc_resize_u8(bgr, h, w, 3, resized, rh, rw, 2 /* INTER_CUBIC */);
  const size_t plane = (size_t)nh * nw;
  for (int y = 0; y < nh; ++y)
    for (int x = 0; x < nw; ++x)
      for (int c = 0; c < 3; ++c) {
        const uint8_t v = (x < rw && y < rh) ? resized[((size_t)y * rw + x) * 3 + c] : 128;  /* copyMakeBorder */
        out[c * plane + (size_t)y * nw + x] = (float)v * (1.0f / 256.0f) + -0.5f;             /* convertTo */
      }
  free(bgr); free(resized);
  return 0;
}
