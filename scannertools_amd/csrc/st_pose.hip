// Pose path of scannertools_caffe (BASELINE config 5): the two deterministic
// ends of the CPM2 / OpenPose COCO-18 pipeline, for gfx950.
//
//   CPM2Input   /root/reference/scannertools_caffe/scannertools_caffe_cpp/cpm2_input_kernel_gpu.cpp:104-140
//               cvtColor(RGB2BGR) -> resize(INTER_CUBIC, frame * scale) -> copyMakeBorder(bottom / right to a
//               multiple of 8, value 128) -> convertTo(F32, 1/256, -0.5) -> split -> planar (3, H, W):
//               SIX passes over the frame in the reference, ONE kernel here (k_cpm2_input): every thread
//               produces one pixel of the network input in all three planes.
//               The arithmetic is that of the OpenCV *CPU* functions of the same names (cv::resize's 8-bit
//               bicubic: A = -0.75, half-pixel centres, 11-bit fixed-point taps -- the same restatement as the
//               Resize op); the reference file calls their cv::cuda twins, whose bicubic is a different filter
//               (A = -0.5, no half-pixel shift), and like the other --build-cuda wrappers it is not the parity
//               target.
//   CPM2Output  .../cpm2_output_kernel_cpu.cpp:424-487 (connect_limbs_coco, candidate scoring): for every limb
//               of the COCO model and every pair of candidate joints, ten samples of the part-affinity field
//               along the segment.  The heat maps are 57 planes of the network's input size (55 MB per 1080p
//               frame at scale 0.34) and live where the network wrote them; k_cpm2_limb_scores samples them
//               in place and hands back max_peaks^2 floats per limb, so the maps never cross PCIe.  The greedy
//               assembly of people from the scored pairs is sequential host work (kernel class).
#include <cmath>
#include <cstring>

#include "st_internal.h"

namespace {

// ---- CPM2Input ------------------------------------------------------------------------------------
struct Cpm2InArgs {
  const uint8_t* const* src;  // n RGB frames (sh, sw, 3)
  float* const* dst;          // n planar (3, nh, nw) float frames
  int sh, sw, rh, rw, nh, nw;
  double scale_x, scale_y;    // source / resized size ratios as cv::resize computes them
};

__device__ __forceinline__ int p_coef(float v) {  // saturate_cast<short>(float): cvRound + saturation
  const float r = rintf(v);
  return r < -32768.f ? -32768 : (r > 32767.f ? 32767 : (int)r);
}
__device__ __forceinline__ void p_cubic(float x, float* c) {  // cv::interpolateCubic, A = -0.75
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

__global__ __launch_bounds__(256) void k_cpm2_input(Cpm2InArgs a) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= a.nw) return;
  const uint8_t* __restrict__ src = a.src[blockIdx.z];
  float* __restrict__ dst = a.dst[blockIdx.z];
  const size_t plane = (size_t)a.nh * a.nw, o = (size_t)y * a.nw + x;
  if (x >= a.rw || y >= a.rh) {  // copyMakeBorder value 128: 128 / 256 - 0.5
    dst[o] = dst[plane + o] = dst[2 * plane + o] = 128 * (1.0f / 256.0f) + -0.5f;
    return;
  }
  int v8[3];
  if (a.rw == a.sw && a.rh == a.sh) {  // cv::resize copies when the sizes are equal
#pragma unroll
    for (int c = 0; c < 3; ++c) v8[c] = src[((size_t)y * a.sw + x) * 3 + c];
  } else {
    float fx = (float)((x + 0.5) * a.scale_x - 0.5);
    const int sx = (int)floorf(fx);
    fx -= sx;
    float fy = (float)((y + 0.5) * a.scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    float cx[4], cy[4];
    p_cubic(fx, cx);
    p_cubic(fy, cy);
    int ax[4], by[4], xs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ax[k] = p_coef(cx[k] * 2048);
      by[k] = p_coef(cy[k] * 2048);
      const int xx = sx - 1 + k;
      xs[k] = (xx < 0 ? 0 : (xx > a.sw - 1 ? a.sw - 1 : xx)) * 3;
    }
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = sy - 1 + k;
      const uint8_t* S = src + (size_t)(yy < 0 ? 0 : (yy > a.sh - 1 ? a.sh - 1 : yy)) * a.sw * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        acc[c] += (S[xs[0] + c] * ax[0] + S[xs[1] + c] * ax[1] + S[xs[2] + c] * ax[2] + S[xs[3] + c] * ax[3]) * by[k];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int q = (acc[c] + (1 << 21)) >> 22;
      v8[c] = q < 0 ? 0 : (q > 255 ? 255 : q);
    }
  }
  // planes in B, G, R order: the frame is RGB and the reference swaps it to BGR before the split
  dst[o] = (float)v8[2] * (1.0f / 256.0f) + -0.5f;
  dst[plane + o] = (float)v8[1] * (1.0f / 256.0f) + -0.5f;
  dst[2 * plane + o] = (float)v8[0] * (1.0f / 256.0f) + -0.5f;
}

// ---- CPM2Output: limb candidate scores ---------------------------------------------------------------
// COCO_18 model tables (cpm2_output_kernel_cpu.cpp:84-88): joint pair of each of the 19 limbs and the two
// part-affinity planes (x, y) of each
__constant__ int kLimbSeq[38] = {1, 2, 1, 5, 2, 3, 3, 4, 5, 6, 6, 7, 1, 8, 8, 9, 9, 10, 1, 11,
                                 11, 12, 12, 13, 1, 0, 0, 14, 14, 16, 0, 15, 15, 17, 2, 16, 5, 17};
__constant__ int kMapIdx[38] = {31, 32, 39, 40, 33, 34, 35, 36, 41, 42, 43, 44, 19, 20, 21, 22, 23, 24, 25,
                                26, 27, 28, 29, 30, 47, 48, 49, 50, 53, 54, 51, 52, 55, 56, 37, 38, 45, 46};

struct LimbArgs {
  const float* const* heatmaps;  // n x (57, h, w)
  const float* const* peaks;     // n x (parts, max_peaks + 1, 3): row 0 = [count, -, -], rows 1.. = (x, y, score)
  float* scores;                 // n x 19 x max_peaks x max_peaks
  int h, w, max_peaks, min_above;
  float threshold;
};

__global__ __launch_bounds__(256) void k_cpm2_limb_scores(LimbArgs a) {
  const int k = blockIdx.x, f = blockIdx.y;
  const float* __restrict__ hm = a.heatmaps[f];
  const float* __restrict__ pk = a.peaks[f];
  const size_t plane = (size_t)a.h * a.w;
  const float* __restrict__ map_x = hm + (size_t)kMapIdx[2 * k] * plane;
  const float* __restrict__ map_y = hm + (size_t)kMapIdx[2 * k + 1] * plane;
  const int peaks_offset = 3 * (a.max_peaks + 1);
  const float* __restrict__ candA = pk + kLimbSeq[2 * k] * peaks_offset;
  const float* __restrict__ candB = pk + kLimbSeq[2 * k + 1] * peaks_offset;
  int nA = (int)candA[0], nB = (int)candB[0];
  nA = nA < 0 ? 0 : (nA > a.max_peaks ? a.max_peaks : nA);
  nB = nB < 0 ? 0 : (nB > a.max_peaks ? a.max_peaks : nB);
  float* __restrict__ out = a.scores + ((size_t)f * 19 + k) * a.max_peaks * a.max_peaks;
  const int num_inter = 10;
  for (int p = threadIdx.x; p < a.max_peaks * a.max_peaks; p += 256) {
    const int i = p / a.max_peaks + 1, j = p - (i - 1) * a.max_peaks + 1;
    float res = -1.f;
    if (i <= nA && j <= nB) {
      const float s_x = candA[i * 3], s_y = candA[i * 3 + 1];
      const float d_x = candB[j * 3] - candA[i * 3], d_y = candB[j * 3 + 1] - candA[i * 3 + 1];
      const float norm_vec = sqrtf(d_x * d_x + d_y * d_y);
      if (!(norm_vec < 1e-6)) {  // coincident peaks are not connected
        const float vec_x = d_x / norm_vec, vec_y = d_y / norm_vec;
        float sum = 0;
        int count = 0;
        for (int lm = 0; lm < num_inter; lm++) {
          int my = (int)roundf(s_y + lm * d_y / num_inter);
          int mx = (int)roundf(s_x + lm * d_x / num_inter);
          if (mx >= a.w) mx = a.w - 1;
          if (my >= a.h) my = a.h - 1;
          if (mx < 0) mx = 0;  // the reference aborts here (CHECK_GE); peaks inside the map never get there
          if (my < 0) my = 0;
          const int idx = my * a.w + mx;
          const float score = vec_x * map_x[idx] + vec_y * map_y[idx];
          if (score > a.threshold) { sum = sum + score; count++; }
        }
        if (count > a.min_above) res = sum / count;
      }
    }
    out[p] = res;
  }
}

// ---- between the network and CPM2Output: x8 up-sampling of the 57 maps and peak extraction ---------------
// In the reference these are the `resize` (ImResizeLayer, configured by cpm2_kernel.cpp:13-29:
// SetStartScale(1), SetScaleGap(0.1), setTargetDimenions(net input size)) and `nms` layers of the Caffe fork
// the CPM2 op links against; their sources are NOT under /root/reference.  What is restated here is the
// published behaviour of those layers ([EXT], from knowledge of the caffe_rtpose / OpenPose sources; unpinned):
//   resize: per output pixel, source position ((x - off) * src/dst, off = dst/src/2 - 0.5), the 4x4 neighbourhood
//           around int(pos + 1e-5) with indices clamped to the map, Catmull-Rom cubic (-0.5, 1.5, -1.5, 0.5 | 1,
//           -2.5, 2, -0.5 | -0.5, 0, 0.5 | 0, 1) along x for the four rows, then along y; one scale (num = 1);
//   nms:    an interior pixel is a peak if it exceeds the threshold and its 8 neighbours; peaks of a part in
//           raster order, the first max_peaks kept: row 0 = [count, 0, 0], row i = (x, y, score).
// Both are HBM-trivial next to the network (55 MB of maps per 1080p frame written once, read once).
constexpr int RM_ROWS = 16;  // output rows per thread: the four row interpolants are reused while the source row stays

struct ResizeMapsArgs {
  const float* src;   // (n, sh, sw, stride) float32, channel-last
  float* const* dst;  // n x (nmaps, th, tw)
  int sh, sw, stride, nmaps, th, tw;
  float off_x, off_y, rx, ry;
  int chan[64];       // source channel of output plane c
};

__device__ __forceinline__ float rm_cubic(float v0, float v1, float v2, float v3, float d) {
  return (-0.5f * v0 + 1.5f * v1 - 1.5f * v2 + 0.5f * v3) * d * d * d + (v0 - 2.5f * v1 + 2.f * v2 - 0.5f * v3) * d * d +
         (-0.5f * v0 + 0.5f * v2) * d + v1;
}

__device__ __forceinline__ void rm_taps(float pos, int len, int nei[4]) {
  int c = (int)((double)pos + 1e-5);
  c = c < 0 ? 0 : c;
  nei[1] = c;
  nei[0] = c - 1 < 0 ? c : c - 1;
  nei[2] = c + 1 >= len ? len - 1 : c + 1;
  nei[3] = nei[2] + 1 >= len ? len - 1 : nei[2] + 1;
}

__global__ __launch_bounds__(256) void k_cpm2_resize_maps(ResizeMapsArgs a) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= a.tw) return;
  const int f = blockIdx.z;
  const float x_on = ((float)x - a.off_x) * a.rx;
  int xn[4];
  rm_taps(x_on, a.sw, xn);
  const float dx = x_on - (float)xn[1];
  const float* __restrict__ src = a.src + (size_t)f * a.sh * a.sw * a.stride;
  float* __restrict__ dst = a.dst[f];
  const int y0 = blockIdx.y * RM_ROWS, y1 = min(a.th, y0 + RM_ROWS);
  const int o0 = xn[0] * a.stride, o1 = xn[1] * a.stride, o2 = xn[2] * a.stride, o3 = xn[3] * a.stride;
  for (int c = 0; c < a.nmaps; ++c) {
    const float* __restrict__ sc = src + a.chan[c];
    int cached = -1;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int y = y0; y < y1; ++y) {  // y is uniform over the workgroup, so is the branch below
      const float y_on = ((float)y - a.off_y) * a.ry;
      int yn[4];
      rm_taps(y_on, a.sh, yn);
      if (yn[1] != cached) {
        cached = yn[1];
        const float* r0 = sc + (size_t)yn[0] * a.sw * a.stride;
        const float* r1 = sc + (size_t)yn[1] * a.sw * a.stride;
        const float* r2 = sc + (size_t)yn[2] * a.sw * a.stride;
        const float* r3 = sc + (size_t)yn[3] * a.sw * a.stride;
        t0 = rm_cubic(r0[o0], r0[o1], r0[o2], r0[o3], dx);
        t1 = rm_cubic(r1[o0], r1[o1], r1[o2], r1[o3], dx);
        t2 = rm_cubic(r2[o0], r2[o1], r2[o2], r2[o3], dx);
        t3 = rm_cubic(r3[o0], r3[o1], r3[o2], r3[o3], dx);
      }
      const float dy = y_on - (float)yn[1];
      dst[((size_t)c * a.th + y) * a.tw + x] = rm_cubic(t0, t1, t2, t3, dy);
    }
  }
}

// Several network scales merged into one set of maps (the `num` loop of the fork's resize kernel; OpenPose's
// resizeAndMerge): source s is sampled with ITS ratio (eff_w / tw source pixels per output pixel, neighbours clamped
// to its own extent), the interpolants are summed in scale order and divided by the number of scales.
constexpr int RM_MAX_SCALES = 8;
struct ResizeMergeArgs {
  const float* src[RM_MAX_SCALES];  // (n, sh[s], sw[s], stride)
  int sh[RM_MAX_SCALES], sw[RM_MAX_SCALES];
  float off_x[RM_MAX_SCALES], off_y[RM_MAX_SCALES], rx[RM_MAX_SCALES], ry[RM_MAX_SCALES];
  float* const* dst;
  int stride, nmaps, th, tw, scales;
  int chan[64];
};

__global__ __launch_bounds__(256) void k_cpm2_resize_merge(ResizeMergeArgs a) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= a.tw) return;
  const int f = blockIdx.z;
  float* __restrict__ dst = a.dst[f];
  const int y0 = blockIdx.y * RM_ROWS, y1 = min(a.th, y0 + RM_ROWS);
  int o[RM_MAX_SCALES][4];
  float dx[RM_MAX_SCALES];
#pragma unroll
  for (int s = 0; s < RM_MAX_SCALES; ++s) {
    if (s < a.scales) {
      const float x_on = ((float)x - a.off_x[s]) * a.rx[s];
      int xn[4];
      rm_taps(x_on, a.sw[s], xn);
      dx[s] = x_on - (float)xn[1];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[s][k] = xn[k] * a.stride;
    }
  }
  for (int c = 0; c < a.nmaps; ++c) {
    int cached[RM_MAX_SCALES];
    float t[RM_MAX_SCALES][4];
#pragma unroll
    for (int s = 0; s < RM_MAX_SCALES; ++s) cached[s] = -1;
    for (int y = y0; y < y1; ++y) {
      float sum = 0.f;
#pragma unroll
      for (int s = 0; s < RM_MAX_SCALES; ++s) {
        if (s < a.scales) {
          const float y_on = ((float)y - a.off_y[s]) * a.ry[s];
          int yn[4];
          rm_taps(y_on, a.sh[s], yn);
          if (yn[1] != cached[s]) {
            cached[s] = yn[1];
            const float* __restrict__ sc = a.src[s] + (size_t)f * a.sh[s] * a.sw[s] * a.stride + a.chan[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float* r = sc + (size_t)yn[k] * a.sw[s] * a.stride;
              t[s][k] = rm_cubic(r[o[s][0]], r[o[s][1]], r[o[s][2]], r[o[s][3]], dx[s]);
            }
          }
          const float v = rm_cubic(t[s][0], t[s][1], t[s][2], t[s][3], y_on - (float)yn[1]);
          sum = s == 0 ? v : sum + v;  // starts from the first interpolant itself: one scale reproduces st_cpm2_resize_maps' bits
        }
      }
      dst[((size_t)c * a.th + y) * a.tw + x] = sum / (float)a.scales;
    }
  }
}

struct NmsArgs {
  const float* const* maps;  // n x (>= parts, h, w)
  float* const* joints;      // n x (parts, max_peaks + 1, 3)
  int h, w, parts, max_peaks;
  float threshold;
};

// one workgroup per (part, frame): the plane in raster order, 1024 pixels per round, ordered compaction
__global__ __launch_bounds__(256) void k_cpm2_nms(NmsArgs a) {
  __shared__ int wave_cnt[4];
  const int part = blockIdx.x, f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* __restrict__ m = a.maps[f] + (size_t)part * a.h * a.w;
  float* __restrict__ out = a.joints[f] + (size_t)part * (a.max_peaks + 1) * 3;
  for (int i = tid; i < (a.max_peaks + 1) * 3; i += 256) out[i] = 0.f;
  __syncthreads();
  const int total = a.h * a.w;
  int count = 0;  // peaks found so far (uniform)
  for (int base = 0; base < total && count < a.max_peaks; base += 1024) {
    unsigned flags = 0;
    int mine = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = base + 4 * tid + k;
      if (idx < total) {
        const int y = idx / a.w, x = idx - y * a.w;
        if (x > 0 && x < a.w - 1 && y > 0 && y < a.h - 1) {
          const float v = m[idx];
          if (v > a.threshold) {
            const float* __restrict__ up = m + idx - a.w;
            const float* __restrict__ dn = m + idx + a.w;
            if (v > up[-1] && v > up[0] && v > up[1] && v > m[idx - 1] && v > m[idx + 1] && v > dn[-1] && v > dn[0] && v > dn[1]) {
              flags |= 1u << k;
              ++mine;
            }
          }
        }
      }
    }
    // exclusive scan of `mine` over the workgroup: shuffles inside a wave, LDS across the four waves
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wave_cnt[wv] = incl;
    __syncthreads();
    int before = count + incl - mine, round_total = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = wave_cnt[q];
      if (q < wv) before += c;
      round_total += c;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (flags & (1u << k)) {
        if (before < a.max_peaks) {
          const int idx = base + 4 * tid + k;
          const int y = idx / a.w, x = idx - y * a.w;
          out[(before + 1) * 3] = (float)x;
          out[(before + 1) * 3 + 1] = (float)y;
          out[(before + 1) * 3 + 2] = m[idx];
        }
        ++before;
      }
    count += round_total;
  }
  if (tid == 0) out[0] = (float)(count < a.max_peaks ? count : a.max_peaks);
}

}  // namespace

ST_EXPORT int st_cpm2_geometry(int h, int w, float scale, int* resize_h, int* resize_w, int* net_h, int* net_w) {
  if (h <= 0 || w <= 0 || !(scale > 0)) return ST_ERR_INVALID;
  // cpm2_input_kernel_gpu.cpp:48-55: i32 = i32 * f32 (float product, truncated), padding to a multiple of 8
  const int rw = (int)(w * scale), rh = (int)(h * scale);
  if (rw <= 0 || rh <= 0) return ST_ERR_INVALID;
  const int wp = (rw % 8) ? 8 - (rw % 8) : 0, hp = (rh % 8) ? 8 - (rh % 8) : 0;
  if (resize_h) *resize_h = rh;
  if (resize_w) *resize_w = rw;
  if (net_h) *net_h = rh + hp;
  if (net_w) *net_w = rw + wp;
  return ST_OK;
}

ST_EXPORT int st_cpm2_scale_for_height(int h, int target_h, float* scale) {
  if (h <= 0 || target_h <= 0 || !scale) return ST_ERR_INVALID;
  float s = (float)target_h / (float)h;
  for (int i = 0; i < 8 && (int)(h * s) < target_h; ++i) s = nextafterf(s, INFINITY);
  for (int i = 0; i < 8 && (int)(h * s) > target_h; ++i) s = nextafterf(s, 0.f);
  if ((int)(h * s) != target_h) return ST_ERR_INVALID;
  *scale = s;
  return ST_OK;
}

ST_EXPORT int st_cpm2_input_batch(st_ctx* ctx, const uint8_t* const* frames_dev, int n, int h, int w, float scale,
                                  float* const* out_dev) {
  ST_TRY(st_enter(ctx));
  int rh, rw, nh, nw;
  if (n < 0 || st_cpm2_geometry(h, w, scale, &rh, &rw, &nh, &nw) != ST_OK || (long long)h * w > 200000000LL ||
      (long long)nh * nw > 200000000LL)
    return st_set_error(ctx, ST_ERR_INVALID, "cpm2_input: bad arguments (n=%d h=%d w=%d scale=%g)", n, h, w, (double)scale);
  if (nh > 65535) return st_set_error(ctx, ST_ERR_UNSUPPORTED, "cpm2_input: network input taller than 65535 rows");
  if (n == 0) return ST_OK;
  if (!frames_dev || !out_dev) return st_set_error(ctx, ST_ERR_INVALID, "cpm2_input: null argument");
  for (int i = 0; i < n; ++i)
    if (!frames_dev[i] || !out_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "cpm2_input: row %d is null", i);
  const size_t tb = st_align_up(sizeof(void*) * (size_t)n);
  ST_TRY(st_ws_reserve(ctx, 2 * tb));
  const uint8_t** d_src = (const uint8_t**)st_ws_alloc(ctx, tb);
  float** d_dst = (float**)st_ws_alloc(ctx, tb);
  ST_HIP(ctx, hipMemcpyAsync(d_src, frames_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  ST_HIP(ctx, hipMemcpyAsync(d_dst, out_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  Cpm2InArgs a;
  a.sh = h; a.sw = w; a.rh = rh; a.rw = rw; a.nh = nh; a.nw = nw;
  a.scale_x = 1. / ((double)rw / w);
  a.scale_y = 1. / ((double)rh / h);
  for (int f0 = 0; f0 < n; f0 += 65535) {
    const int nf = n - f0 < 65535 ? n - f0 : 65535;
    a.src = d_src + f0; a.dst = d_dst + f0;
    st_timed t(ctx, ST_K_CPM2_INPUT);
    hipLaunchKernelGGL(k_cpm2_input, dim3((nw + 255) / 256, nh, nf), dim3(256), 0, ctx->stream, a);
    ST_HIP(ctx, hipGetLastError());
  }
  return ST_OK;
}

ST_EXPORT int st_cpm2_limb_scores(st_ctx* ctx, const float* const* heatmaps_dev, const float* const* peaks_dev, int n,
                                  int net_h, int net_w, int max_peaks, float inter_threshold, int min_above,
                                  float* scores_dev) {
  ST_TRY(st_enter(ctx));
  if (n < 0 || net_h <= 0 || net_w <= 0 || max_peaks < 1 || max_peaks > 1024 || (long long)net_h * net_w > 35000000LL)
    return st_set_error(ctx, ST_ERR_INVALID, "cpm2_limb_scores: bad arguments (n=%d %dx%d max_peaks=%d)", n, net_h, net_w, max_peaks);
  if (n == 0) return ST_OK;
  if (!heatmaps_dev || !peaks_dev || !scores_dev) return st_set_error(ctx, ST_ERR_INVALID, "cpm2_limb_scores: null argument");
  for (int i = 0; i < n; ++i)
    if (!heatmaps_dev[i] || !peaks_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "cpm2_limb_scores: row %d is null", i);
  const size_t tb = st_align_up(sizeof(void*) * (size_t)n);
  ST_TRY(st_ws_reserve(ctx, 2 * tb));
  const float** d_hm = (const float**)st_ws_alloc(ctx, tb);
  const float** d_pk = (const float**)st_ws_alloc(ctx, tb);
  ST_HIP(ctx, hipMemcpyAsync(d_hm, heatmaps_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  ST_HIP(ctx, hipMemcpyAsync(d_pk, peaks_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  LimbArgs a;
  a.h = net_h; a.w = net_w; a.max_peaks = max_peaks; a.min_above = min_above; a.threshold = inter_threshold;
  for (int f0 = 0; f0 < n; f0 += 65535) {
    const int nf = n - f0 < 65535 ? n - f0 : 65535;
    a.heatmaps = d_hm + f0; a.peaks = d_pk + f0;
    a.scores = scores_dev + (size_t)f0 * 19 * max_peaks * max_peaks;
    st_timed t(ctx, ST_K_CPM2_LIMBS);
    hipLaunchKernelGGL(k_cpm2_limb_scores, dim3(19, nf), dim3(256), 0, ctx->stream, a);
    ST_HIP(ctx, hipGetLastError());
  }
  return ST_OK;
}

ST_EXPORT int st_cpm2_resize_maps(st_ctx* ctx, const float* src_dev, int n, int src_h, int src_w, int src_stride,
                                  const int* chan_map, int nmaps, int dst_h, int dst_w, float* const* out_dev) {
  ST_TRY(st_enter(ctx));
  if (n < 0 || src_h <= 0 || src_w <= 0 || src_stride <= 0 || nmaps <= 0 || nmaps > 64 || dst_h <= 0 || dst_w <= 0 ||
      (long long)dst_h * dst_w > 35000000LL || n > 65535)
    return st_set_error(ctx, ST_ERR_INVALID, "cpm2_resize_maps: bad arguments (n=%d %dx%d -> %dx%d, %d maps)", n, src_h, src_w, dst_h, dst_w, nmaps);
  if (n == 0) return ST_OK;
  if (!src_dev || !out_dev) return st_set_error(ctx, ST_ERR_INVALID, "cpm2_resize_maps: null argument");
  ResizeMapsArgs a;
  for (int c = 0; c < nmaps; ++c) {
    a.chan[c] = chan_map ? chan_map[c] : c;
    if (a.chan[c] < 0 || a.chan[c] >= src_stride) return st_set_error(ctx, ST_ERR_INVALID, "cpm2_resize_maps: channel %d outside the source pixel", a.chan[c]);
  }
  for (int i = 0; i < n; ++i)
    if (!out_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "cpm2_resize_maps: row %d is null", i);
  const size_t tb = st_align_up(sizeof(void*) * (size_t)n);
  ST_TRY(st_ws_reserve(ctx, tb));
  float** d_dst = (float**)st_ws_alloc(ctx, tb);
  ST_HIP(ctx, hipMemcpyAsync(d_dst, out_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  a.src = src_dev; a.dst = d_dst;
  a.sh = src_h; a.sw = src_w; a.stride = src_stride; a.nmaps = nmaps; a.th = dst_h; a.tw = dst_w;
  // the layer's own arithmetic: `tw/float(ow)/2 - 0.5` is a float quotient minus a double constant, stored as float
  a.off_x = (float)((double)((float)dst_w / (float)src_w / 2) - 0.5);
  a.off_y = (float)((double)((float)dst_h / (float)src_h / 2) - 0.5);
  a.rx = (float)src_w / (float)dst_w;
  a.ry = (float)src_h / (float)dst_h;
  st_timed t(ctx, ST_K_CPM2_RESIZE);
  hipLaunchKernelGGL(k_cpm2_resize_maps, dim3((dst_w + 255) / 256, (dst_h + RM_ROWS - 1) / RM_ROWS, n), dim3(256), 0, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

ST_EXPORT int st_cpm2_resize_merge_maps(st_ctx* ctx, const float* const* src_dev, const int* src_h, const int* src_w, const float* eff_h,
                                        const float* eff_w, int scales, int n, int src_stride, const int* chan_map, int nmaps, int dst_h,
                                        int dst_w, float* const* out_dev) {
  ST_TRY(st_enter(ctx));
  if (scales < 1 || scales > RM_MAX_SCALES || n < 0 || src_stride <= 0 || nmaps <= 0 || nmaps > 64 || dst_h <= 0 || dst_w <= 0 ||
      (long long)dst_h * dst_w > 35000000LL || n > 65535 || !src_h || !src_w || !eff_h || !eff_w)
    return st_set_error(ctx, ST_ERR_INVALID, "cpm2_resize_merge_maps: bad arguments (%d scales, n=%d -> %dx%d, %d maps)", scales, n, dst_h, dst_w, nmaps);
  if (n == 0) return ST_OK;
  if (!src_dev || !out_dev) return st_set_error(ctx, ST_ERR_INVALID, "cpm2_resize_merge_maps: null argument");
  ResizeMergeArgs a;
  memset(&a, 0, sizeof(a));
  for (int c = 0; c < nmaps; ++c) {
    a.chan[c] = chan_map ? chan_map[c] : c;
    if (a.chan[c] < 0 || a.chan[c] >= src_stride) return st_set_error(ctx, ST_ERR_INVALID, "cpm2_resize_merge_maps: channel %d outside the source pixel", a.chan[c]);
  }
  for (int s = 0; s < scales; ++s) {
    if (!src_dev[s] || src_h[s] <= 0 || src_w[s] <= 0 || !(eff_h[s] > 0.f) || !(eff_w[s] > 0.f))
      return st_set_error(ctx, ST_ERR_INVALID, "cpm2_resize_merge_maps: scale %d has no maps or an empty extent", s);
    a.src[s] = src_dev[s]; a.sh[s] = src_h[s]; a.sw[s] = src_w[s];
    // st_cpm2_resize_maps' arithmetic with the scale's effective extent in place of the map size
    a.off_x[s] = (float)((double)((float)dst_w / eff_w[s] / 2) - 0.5);
    a.off_y[s] = (float)((double)((float)dst_h / eff_h[s] / 2) - 0.5);
    a.rx[s] = eff_w[s] / (float)dst_w;
    a.ry[s] = eff_h[s] / (float)dst_h;
  }
  for (int i = 0; i < n; ++i)
    if (!out_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "cpm2_resize_merge_maps: row %d is null", i);
  const size_t tb = st_align_up(sizeof(void*) * (size_t)n);
  ST_TRY(st_ws_reserve(ctx, tb));
  float** d_dst = (float**)st_ws_alloc(ctx, tb);
  ST_HIP(ctx, hipMemcpyAsync(d_dst, out_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  a.dst = d_dst; a.stride = src_stride; a.nmaps = nmaps; a.th = dst_h; a.tw = dst_w; a.scales = scales;
  st_timed t(ctx, ST_K_CPM2_RESIZE);
  hipLaunchKernelGGL(k_cpm2_resize_merge, dim3((dst_w + 255) / 256, (dst_h + RM_ROWS - 1) / RM_ROWS, n), dim3(256), 0, ctx->stream, a);
  ST_HIP(ctx, hipGetLastError());
  return ST_OK;
}

ST_EXPORT int st_cpm2_nms(st_ctx* ctx, const float* const* maps_dev, int n, int h, int w, int parts, int max_peaks,
                          float threshold, float* const* joints_dev) {
  ST_TRY(st_enter(ctx));
  if (n < 0 || h <= 0 || w <= 0 || parts <= 0 || parts > 65535 || max_peaks < 1 || max_peaks > 1024 || (long long)h * w > 35000000LL)
    return st_set_error(ctx, ST_ERR_INVALID, "cpm2_nms: bad arguments (n=%d %dx%d parts=%d max_peaks=%d)", n, h, w, parts, max_peaks);
  if (n == 0) return ST_OK;
  if (!maps_dev || !joints_dev) return st_set_error(ctx, ST_ERR_INVALID, "cpm2_nms: null argument");
  for (int i = 0; i < n; ++i)
    if (!maps_dev[i] || !joints_dev[i]) return st_set_error(ctx, ST_ERR_INVALID, "cpm2_nms: row %d is null", i);
  const size_t tb = st_align_up(sizeof(void*) * (size_t)n);
  ST_TRY(st_ws_reserve(ctx, 2 * tb));
  const float** d_maps = (const float**)st_ws_alloc(ctx, tb);
  float** d_j = (float**)st_ws_alloc(ctx, tb);
  ST_HIP(ctx, hipMemcpyAsync(d_maps, maps_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  ST_HIP(ctx, hipMemcpyAsync(d_j, joints_dev, sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  NmsArgs a;
  a.h = h; a.w = w; a.parts = parts; a.max_peaks = max_peaks; a.threshold = threshold;
  for (int f0 = 0; f0 < n; f0 += 65535) {
    const int nf = n - f0 < 65535 ? n - f0 : 65535;
    a.maps = d_maps + f0; a.joints = d_j + f0;
    st_timed t(ctx, ST_K_CPM2_NMS);
    hipLaunchKernelGGL(k_cpm2_nms, dim3(parts, nf), dim3(256), 0, ctx->stream, a);
    ST_HIP(ctx, hipGetLastError());
  }
  return ST_OK;
}
