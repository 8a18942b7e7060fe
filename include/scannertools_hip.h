/*
 * scannertools_hip.h -- C ABI of libscannertools_hip.so: the MI355X (gfx950) implementation of
 * the arithmetic behind scannertools' Histogram and OpticalFlow ops.
 *
 * This is the boundary between host code (the Scanner kernels in
 * the scannertools_amd/scanner_kernels sources, the Python front-end, bench.py) and the HIP
 * kernels.  Plain pointers and sizes only; no C++ or torch types.  Every entry point
 *   - returns an int status (ST_OK == 0), never throws, never aborts;
 *   - selects the context's device on entry (Scanner calls kernel instances from their own
 *     evaluator threads; cf. set_device() at the top of every method in
 *     scannertools_cpp/imgproc/histogram_kernel_gpu.cpp:21,35 and
 *     optical_flow_kernel_gpu.cpp:18,29,37,41,47);
 *   - never allocates outputs: the caller passes device buffers it owns (Scanner allocates
 *     them with new_block_buffer / new_frames, histogram_kernel_gpu.cpp:38-39,
 *     optical_flow_kernel_gpu.cpp:61-64);
 *   - enqueues work on the context's stream and returns without synchronising, unless
 *     documented otherwise; call st_ctx_sync() before reading results on the host;
 *   - is re-entrant across contexts (one context per kernel instance / thread).
 *
 * "frames_dev" arguments are HOST arrays of DEVICE pointers (one Scanner element buffer per
 * frame); frames are dense interleaved U8 (h, w, 3), RGB order, no row padding -- the layout
 * frame_to_mat()/frame_to_gpu_mat() view in the reference.
 */
#ifndef SCANNERTOOLS_HIP_H_
#define SCANNERTOOLS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ST_ABI_VERSION 1

enum st_status {
  ST_OK = 0,
  ST_ERR_INVALID = 1,     /* bad argument (null pointer, non-positive size, bins out of range ...) */
  ST_ERR_HIP = 2,         /* a HIP runtime call failed; see st_ctx_last_error() */
  ST_ERR_OOM = 3,         /* workspace allocation failed */
  ST_ERR_UNSUPPORTED = 4  /* parameter combination the HIP path does not implement */
};

typedef struct st_ctx st_ctx;

int st_abi_version(void);
/* "src=<first 16 hex digits of the sha256 over the library's sources, in the Makefile's order> host=<machine that compiled
 * it> at=<UTC time>": which sources this binary was built from and where (no reference counterpart: build provenance, so
 * that a test run can show it loaded a library compiled on the machine it runs on from the tree it runs in). */
const char* st_build_info(void);
const char* st_status_string(int status);
int st_device_count(int* count);

/* Per-kernel-instance context: owns a stream and lazily-sized scratch (pyramids, polynomial
 * expansions, matrices).  Replaces the per-instance state of the reference GPU wrappers
 * (streams_, planes_, flow_finders_, grayscale_; histogram_kernel_gpu.cpp:71-75,
 * optical_flow_kernel_gpu.cpp:101-107). */
/* ST_ERR_UNSUPPORTED: the device is not a gfx950-class part (the kernels are compiled for gfx950 and sized for its 160 KB
 * of LDS per workgroup); ST_ERR_INVALID: no such device. */
int st_ctx_create(int device_id, st_ctx** out_ctx);
int st_ctx_destroy(st_ctx* ctx);
/* Borrow an external hipStream_t (e.g. torch's current stream).  NULL is the HIP default
 * (null) stream, which is what torch's default stream is.  st_ctx_reset_stream() returns to
 * the context's own non-blocking stream.  Both drain the previously bound stream first. */
int st_ctx_set_stream(st_ctx* ctx, void* hip_stream);
int st_ctx_reset_stream(st_ctx* ctx);
int st_ctx_sync(st_ctx* ctx);
/* Diagnostic: 1 if the context's last st_farneback_pairs call chose its kernels for a GPU shared with other kernel instances of
 * this process (two or more other contexts had an OpticalFlow call in flight -- entered within 50 ms and not yet st_ctx_sync'ed --
 * and the call had at most 4 pairs), else 0.  Scanner runs K instances of a kernel class per GPU (pipeline_instances_per_node,
 * scannertools/tests/test_all.py:45,231); the choice is a scheduling matter only, results are bit-identical.  ST_CONCURRENT=0 / 1
 * (read at st_ctx_create) switches the detection off / forces it. */
int st_ctx_flow_concurrent(st_ctx* ctx);
/* Cap on scratch the context may hold (bytes; 0 = default 64 GiB).  Large pair batches are
 * processed in passes that fit. */
int st_ctx_set_workspace_limit(st_ctx* ctx, size_t bytes);
int st_ctx_release_workspace(st_ctx* ctx);
const char* st_ctx_last_error(const st_ctx* ctx);

/* ---- per-kernel timing (HIP events on the context's stream) --------------------------------
 * When enabled, every launch of the named kernel class is bracketed by hipEventRecord pairs;
 * st_ctx_timing_read() synchronises the stream and returns launches and total milliseconds
 * since the last reset.  Used by bench.py for the live roofline figure. */
enum st_kernel_id {
  ST_K_HIST = 0,
  ST_K_GRAY = 1,
  ST_K_PYR = 2,
  ST_K_POLYEXP = 3,
  ST_K_UPDATE_MATRICES = 4,
  ST_K_BLUR_UPDATE = 5, /* box blur + 2x2 solve (+ fused UpdateMatrices): the dominant kernel */
  ST_K_FLOW_HIST = 6,
  ST_K_DRAW_FLOW = 7,   /* max-reduction + render launches of one st_draw_flow_batch call */
  ST_K_BLUR_OP = 8,     /* the Blur op's box filter (not the Farneback blur, which is ST_K_BLUR_UPDATE) */
  ST_K_RESIZE = 9,
  ST_K_CVT_COLOR = 10,
  ST_K_CPM2_INPUT = 11,
  ST_K_CPM2_LIMBS = 12,
  ST_K_CONV = 13,       /* convolution / pooling launches of the pose network */
  ST_K_CPM2_RESIZE = 14,
  ST_K_CPM2_NMS = 15,
  ST_K_COUNT = 16
};
int st_ctx_timing_enable(st_ctx* ctx, unsigned kernel_mask);
int st_ctx_timing_reset(st_ctx* ctx);
int st_ctx_timing_read(st_ctx* ctx, int kernel_id, int* launches, double* total_ms);

/* ---- Histogram -----------------------------------------------------------------------------
 * Replaces HistogramKernelCPU::execute's per-frame body
 * (scannertools_cpp/imgproc/histogram_kernel_cpu.cpp:25-45: cv::calcHist x3 + convertTo(CV_32S))
 * and the GPU wrapper's cvc::split + cvc::histEven x3 (histogram_kernel_gpu.cpp:48-57) for a
 * whole batch in one call.
 * out_dev: n * 3 * bins int32, frame-major then channel-major (R,G,B) -- element i is the
 * 3*bins*4-byte block the reference hands to insert_element() (histogram_kernel_cpu.cpp:44).
 * bins in [1, 256]; bin(v) = floor(v * bins / 256) (cv::calcHist uniform 8U table); the
 * reference's value is 16 (histogram_kernel_cpu.cpp:8). */
int st_hist_u8c3_batch(st_ctx* ctx, const uint8_t* const* frames_dev, int n, int h, int w,
                       int bins, int32_t* out_dev);
/* Same, frames at base_dev + i * frame_stride_bytes (a contiguous device-resident stream). */
int st_hist_u8c3_strided(st_ctx* ctx, const uint8_t* base_dev, size_t frame_stride_bytes, int n,
                         int h, int w, int bins, int32_t* out_dev);

/* ---- ShotBoundaries (device-resident histograms) -------------------------------------------
 * Replaces the body of the ShotBoundaries python op, scannertools/shot_detection.py:12-28, for histograms that are
 * already on the GPU (the op itself stays host code in the reference; with resident frames its 10 000-window loop is
 * two thirds of the Histogram -> ShotBoundaries pipeline).  hist_dev: n * 3 * bins int32 as st_hist_u8c3_* writes
 * them; flags_dev[i] = 1 where frame i opens a shot: diffs[i] - mean(win) > k_std * std(win), win =
 * diffs[max(i - window, 0) : min(i + window, n)], diffs[i] = mean over the channels of the Chebyshev distance between
 * frames i-1 and i (shot_detection.py:14-18, :23-26; the reference's window = 500, k_std = 2.5).  Means and standard
 * deviations are formed in numpy's summation order, so the flags equal the reference's decisions bit for bit.
 * diffs_dev: optional n doubles that receive diffs (null: context scratch). */
int st_shot_boundaries(st_ctx* ctx, const int32_t* hist_dev, int n, int bins, int window, double k_std, uint8_t* flags_dev,
                       double* diffs_dev);

/* ---- OpticalFlow ---------------------------------------------------------------------------
 * Parameters of cv::FarnebackOpticalFlow::create(numLevels, pyrScale, fastPyramids, winSize,
 * numIters, polyN, polySigma, flags); st_fb_params_default() gives the reference's
 * (3, 0.5, false, 15, 3, 5, 1.2, 0) (optical_flow_kernel_cpu.cpp:15-16).  Implemented:
 * flags == 0 (box filter, no initial flow), fast_pyramids == 0, poly_n in {5, 7},
 * odd win_size <= 63.  gray_bits selects cv::cvtColor's 8-bit luma table: 15 (OpenCV 4.x)
 * or 14 (OpenCV <= 3.4.2). */
typedef struct st_fb_params {
  int num_levels;
  double pyr_scale;
  int fast_pyramids;
  int win_size;
  int num_iters;
  int poly_n;
  double poly_sigma;
  int flags;
  int gray_bits;
} st_fb_params;
void st_fb_params_default(st_fb_params* p);

/* Replaces OpticalFlowKernelCPU::execute (optical_flow_kernel_cpu.cpp:27-43: cvtColor
 * BGR2GRAY x2 + FarnebackOpticalFlow::calc) for a batch of stencil windows, and the
 * calling convention of OpticalFlowKernelGPU::execute (optical_flow_kernel_gpu.cpp:45-93).
 * pairs: HOST array of n_pairs x 2 indices into frames_dev; flow p goes FROM frame
 * pairs[2p] TO frame pairs[2p+1] (CPU-kernel direction: stencil element 0 -> element 1;
 * the reference GPU wrapper's reversed direction is a defect and is not reproduced).
 * Pairs need not be consecutive; each distinct frame's pyramid and polynomial expansion is
 * computed once per call.  flow_out_dev: HOST array of n_pairs DEVICE pointers, each a dense
 * (h, w, 2) F32 frame (u, v interleaved) as allocated by new_frames(device, FrameInfo(h, w, 2,
 * F32), n) (optical_flow_kernel_gpu.cpp:61-64). */
int st_farneback_pairs(st_ctx* ctx, const uint8_t* const* frames_dev, int n_frames,
                       const int32_t* pairs, int n_pairs, int h, int w, const st_fb_params* params,
                       float* const* flow_out_dev);

/* Number of pyramid levels processed minus one (k runs levels..0) and per-level geometry, as
 * FarnebackOpticalFlowImpl::calc derives them. */
int st_fb_levels(int h, int w, const st_fb_params* params);
int st_fb_level_geom(int h, int w, const st_fb_params* params, int level, int* lh, int* lw,
                     double* sigma, int* ksize);

/* ---- stage-level entry points (parity tests drive each Farneback stage separately) ---------
 * All arrays are dense device arrays.  M fields are planar (5, h, w) F32 (channel c at
 * base + c*h*w); flow fields are (h, w, 2) F32 interleaved.  Polynomial expansions R (r_dev,
 * r0_dev, r1_dev; 5*h*w floats, 16-byte aligned) use the layout the production kernels read:
 * channels 0..3 of OpenCV's interleaved 5-channel R as h*w float4, then channel 4 as h*w
 * floats. */
int st_gray_u8(st_ctx* ctx, const uint8_t* rgb_dev, int h, int w, int gray_bits, uint8_t* gray_dev);
int st_fb_pyr_image(st_ctx* ctx, const uint8_t* gray_dev, int h, int w, const st_fb_params* params,
                    int level, float* img_dev /* (lh, lw) */);
int st_fb_polyexp(st_ctx* ctx, const float* img_dev, int h, int w, int poly_n, double poly_sigma,
                  float* r_dev /* R layout, 5*h*w floats */);
/* M = UpdateMatrices(R0, R1, flow).  If coarse_flow_dev != NULL the flow is first produced as
 * resize(coarse_flow (ch, cw, 2) -> (h, w), INTER_LINEAR) * (1/pyr_scale) (the level
 * transition of calc()); else flow_dev (h, w, 2) is used; if both are NULL the flow is zero. */
int st_fb_update_matrices(st_ctx* ctx, const float* r0_dev, const float* r1_dev, const float* flow_dev,
                          const float* coarse_flow_dev, int ch, int cw, double pyr_scale, int h, int w,
                          float* m_dev /* (5,h,w) */);
/* One FarnebackUpdateFlow_Blur pass: flow_out = solve(box(M)); if update != 0 additionally
 * m_out = UpdateMatrices(R0, R1, flow_out).  m_out must not alias m_in. */
int st_fb_update_flow_blur(st_ctx* ctx, const float* r0_dev, const float* r1_dev, const float* m_in_dev,
                           int h, int w, int block_size, int update, float* flow_out_dev,
                           float* m_out_dev);

/* One whole iteration as the production path runs it (block_size 15 only):
 *   flow_out = solve(box(UpdateMatrices(R0, R1, flow_in)))  without materialising M.
 * flow_in: flow_in_dev (h,w,2), or resize(coarse_flow (ch,cw,2))*(1/pyr_scale), or zero when both
 * are NULL.  Equals st_fb_update_matrices followed by st_fb_update_flow_blur(update = 0). */
int st_fb_flow_iteration(st_ctx* ctx, const float* r0_dev, const float* r1_dev, const float* flow_in_dev,
                         const float* coarse_flow_dev, int ch, int cw, double pyr_scale, int h, int w,
                         int block_size, float* flow_out_dev);

/* ---- Flow consumers ------------------------------------------------
 * FlowHistogram: replaces FlowHistogramKernelCPU::execute's per-frame body
 * (scannertools/old/cpp_ops/flow_histogram_kernel_cpu.cpp:26-57: cv::split, cv::cartToPolar(x, y,
 * mag, deg, true), cv::calcHist on mag over [0,64) and on deg over [0,360), 64 bins each,
 * convertTo(CV_32S)) for a whole batch in one launch.
 * flows: n device pointers to (h, w, 2) float32 flow frames (x then y, the layout OpticalFlow
 * emits), 8-byte aligned.  out_dev: n * 2 * 64 int32, per frame the magnitude row then the angle
 * row -- the 512-byte block the reference hands to insert_element() (:55).  Values outside the
 * ranges (mag >= 64, deg == 360, NaN) are not counted, as in cv::calcHist. */
int st_flow_hist_batch(st_ctx* ctx, const float* const* flows_dev, int n, int h, int w, int32_t* out_dev);
/* Same, flow frames at base_dev + i * frame_stride_bytes. */
int st_flow_hist_strided(st_ctx* ctx, const float* base_dev, size_t frame_stride_bytes, int n, int h,
                         int w, int32_t* out_dev);

/* DrawFlow: replaces draw_flow() of scannertools/vis.py:8-12 for a batch:
 * out = hstack(frame, uint8(min(avg / max(avg), 1) * 255) on 3 channels), avg = (fx + fy) / 2 in
 * float32, max taken over the frame (NaN propagates, as np.max does), numpy's float32 -> uint8
 * cast.  frames: (h, w, 3) uint8; flows: (h, w, 2) float32, 8-byte aligned; out: (h, 2w, 3) uint8. */
int st_draw_flow_batch(st_ctx* ctx, const uint8_t* const* frames_dev, const float* const* flows_dev,
                       int n, int h, int w, uint8_t* const* out_dev);

/* ---- Sibling imgproc ops ---------------------------------------------
 * Blur: replaces BlurKernel::execute (scannertools_cpp/imgproc/blur_kernel_cpu.cpp:50-81), the
 * reference's own box filter: window [-left, +right] around each interior pixel with
 * left = ceil(kernel_size/2.0) - 1, right = kernel_size/2, per channel
 * out = sum / (left + right + 1)^2 in unsigned integer arithmetic; BlurArgs.sigma is ignored, as in
 * the reference.  frames / out: n device pointers to (h, w, 3) uint8; out must not alias its input.
 * Border pixels (which the reference leaves uninitialised) are set to 0.  kernel_size in [1, 31]. */
int st_box_blur_u8c3_batch(st_ctx* ctx, const uint8_t* const* frames_dev, int n, int h, int w,
                           int kernel_size, uint8_t* const* out_dev);

/* Resize: replaces the cv::resize(img, out, Size(out_w, out_h), 0, 0, interpolation) call of
 * ResizeKernel::execute (scannertools_cpp/imgproc/resize_kernel.cpp:68-73) for U8 frames of 1..4
 * channels.  interpolation takes cv::InterpolationFlags values; ST_INTER_LINEAR (the op's default,
 * resize_kernel.cpp:31), ST_INTER_NEAREST, ST_INTER_CUBIC, ST_INTER_AREA and ST_INTER_LANCZOS4 are
 * implemented -- every entry of the reference's table (resize_kernel.cpp:10-17) except the
 * INTER_MAX mask value, which returns ST_ERR_UNSUPPORTED.
 * The target size is the caller's business (ResizeArgs width/height/min/preserve_aspect,
 * resize_kernel.cpp:44-62, is evaluated by the kernel class). */
enum st_interpolation { ST_INTER_NEAREST = 0, ST_INTER_LINEAR = 1, ST_INTER_CUBIC = 2, ST_INTER_AREA = 3, ST_INTER_LANCZOS4 = 4 };
int st_resize_u8_batch(st_ctx* ctx, const uint8_t* const* frames_dev, int n, int h, int w, int channels,
                       int out_h, int out_w, int interpolation, uint8_t* const* out_dev);

/* ConvertColor: replaces the cv::cvtColor(img, out, code) call of ConvertColorKernel::execute
 * (scannertools_cpp/imgproc/convert_color_kernel.cpp:268-271) for U8 frames.  code takes
 * cv::ColorConversionCodes values; implemented: the ones below (the reference's name table at
 * convert_color_kernel.cpp:10-209 lists many more, which return ST_ERR_UNSUPPORTED).  gray_bits
 * selects the luma table width as in st_fb_params (15: OpenCV 4.x, 14: <= 3.4.2).
 * st_cvt_color_out_channels() gives the channel count of the output frame (-1: unsupported). */
enum st_color_code {
  /* channel layout family: codes 0..3, 5, 9..31 (alpha channel added / dropped / swapped, 16-bit BGR565 / BGR555
   * pixels as 2-channel frames, gray from / to them) */
  ST_COLOR_BGR2BGRA = 0, ST_COLOR_BGRA2BGR = 1, ST_COLOR_BGR2RGBA = 2, ST_COLOR_RGBA2BGR = 3, ST_COLOR_BGRA2RGBA = 5,
  ST_COLOR_GRAY2BGRA = 9, ST_COLOR_BGRA2GRAY = 10, ST_COLOR_RGBA2GRAY = 11,
  ST_COLOR_BGR2BGR565 = 12, ST_COLOR_RGB2BGR565 = 13, ST_COLOR_BGR5652BGR = 14, ST_COLOR_BGR5652RGB = 15,
  ST_COLOR_BGRA2BGR565 = 16, ST_COLOR_RGBA2BGR565 = 17, ST_COLOR_BGR5652BGRA = 18, ST_COLOR_BGR5652RGBA = 19,
  ST_COLOR_GRAY2BGR565 = 20, ST_COLOR_BGR5652GRAY = 21,
  ST_COLOR_BGR2BGR555 = 22, ST_COLOR_RGB2BGR555 = 23, ST_COLOR_BGR5552BGR = 24, ST_COLOR_BGR5552RGB = 25,
  ST_COLOR_BGRA2BGR555 = 26, ST_COLOR_RGBA2BGR555 = 27, ST_COLOR_BGR5552BGRA = 28, ST_COLOR_BGR5552RGBA = 29,
  ST_COLOR_GRAY2BGR555 = 30, ST_COLOR_BGR5552GRAY = 31,
  ST_COLOR_BGR2RGB = 4, ST_COLOR_RGB2BGR = 4,
  ST_COLOR_BGR2GRAY = 6, ST_COLOR_RGB2GRAY = 7,
  ST_COLOR_GRAY2BGR = 8, ST_COLOR_GRAY2RGB = 8,
  ST_COLOR_BGR2XYZ = 32, ST_COLOR_RGB2XYZ = 33, ST_COLOR_XYZ2BGR = 34, ST_COLOR_XYZ2RGB = 35,
  ST_COLOR_BGR2YCrCb = 36, ST_COLOR_RGB2YCrCb = 37, ST_COLOR_YCrCb2BGR = 38, ST_COLOR_YCrCb2RGB = 39,
  ST_COLOR_BGR2HSV = 40, ST_COLOR_RGB2HSV = 41, ST_COLOR_HSV2BGR = 54, ST_COLOR_HSV2RGB = 55,
  ST_COLOR_BGR2HSV_FULL = 66, ST_COLOR_RGB2HSV_FULL = 67, ST_COLOR_HSV2BGR_FULL = 70, ST_COLOR_HSV2RGB_FULL = 71,
  ST_COLOR_BGR2HLS = 52, ST_COLOR_RGB2HLS = 53, ST_COLOR_HLS2BGR = 60, ST_COLOR_HLS2RGB = 61,
  ST_COLOR_BGR2HLS_FULL = 68, ST_COLOR_RGB2HLS_FULL = 69, ST_COLOR_HLS2BGR_FULL = 72, ST_COLOR_HLS2RGB_FULL = 73,
  ST_COLOR_BGR2YUV = 82, ST_COLOR_RGB2YUV = 83, ST_COLOR_YUV2BGR = 84, ST_COLOR_YUV2RGB = 85
};
int st_cvt_color_out_channels(int code, int in_channels);
/* Output shape of a conversion: most codes keep (h, w) and change the channel count (st_cvt_color_out_channels);
 * the YUV 4:2:0 sources (cv::cvtColor codes 90..106: NV12, NV21, YV12, IYUV / I420 -- a (3H/2, W) single-channel frame, what a
 * video decoder hands out) produce (H, W, 3 | 4 | 1); the packed 4:2:2 sources (107..124: UYVY, YUY2, YVYU; (H, W, 2))
 * produce (H, W, 3 | 4 | 1).  Returns 0, or -1 when the code does not apply to such a frame.  The output-shape probe of
 * ConvertColorKernel::execute (convert_color_kernel.cpp:252-277). */
int st_cvt_color_out_shape(int code, int in_h, int in_w, int in_channels, int* out_h, int* out_w, int* out_channels);
int st_cvt_color_u8_batch(st_ctx* ctx, const uint8_t* const* frames_dev, int n, int h, int w, int channels,
                          int code, int gray_bits, uint8_t* const* out_dev);

/* ---- Pose path (scannertools_caffe) ------------------------------------
 * Geometry of the CPM2 network input for a frame of (h, w) at `scale`, as CPM2InputKernel::new_frame_info
 * and CPM2OutputKernel::new_frame_info derive it (scannertools_caffe_cpp/cpm2_input_kernel_gpu.cpp:44-55,
 * cpm2_output_kernel_cpu.cpp:123-137): resize = (int)(size * scale) in float, padded up to a multiple of 8.
 * Host-only (no context). */
int st_cpm2_geometry(int h, int w, float scale, int* resize_h, int* resize_w, int* net_h, int* net_w);

/* The float scale for which the rule above resizes a frame of height h to EXACTLY target_h rows: (int)(h * scale)
 * == target_h.  (float)target_h / h can fall one row short under the truncation -- 368.f / 1080 gives 367 -- which
 * would add a padded row and shift every normalised y coordinate of the OpenPose op, whose network input height is a
 * given (op::Wrapper netInputSize = (-1, 368), scannertools_caffe_cpp/openpose_kernel.cpp:99).  Host-only. */
int st_cpm2_scale_for_height(int h, int target_h, float* scale);

/* CPM2Input: replaces the per-frame body of CPM2InputKernel::execute
 * (cpm2_input_kernel_gpu.cpp:104-140: cvtColor RGB2BGR, resize INTER_CUBIC, copyMakeBorder with 128,
 * convertTo(F32, 1/256, -0.5), split, three plane copies, cudaMemcpy2DAsync) for a whole batch in one
 * launch.  frames: n device pointers to (h, w, 3) uint8 RGB frames; out: n device pointers to dense
 * planar (3, net_h, net_w) float32 frames (planes B, G, R), the FrameInfo(3, net_h, net_w, F32) frame the
 * reference hands to insert_frame (:96-103).  Arithmetic: OpenCV's CPU functions of the same names (the
 * reference file calls their cv::cuda twins, whose bicubic is a different filter; see st_pose.hip). */
int st_cpm2_input_batch(st_ctx* ctx, const uint8_t* const* frames_dev, int n, int h, int w, float scale,
                        float* const* out_dev);

/* CPM2Output, candidate scoring: replaces the pair loop of CPM2OutputKernel::connect_limbs_coco
 * (cpm2_output_kernel_cpu.cpp:424-487) for all 19 limbs of the COCO_18 model and a batch of frames.
 * heatmaps: n device pointers to (57, net_h, net_w) float32 maps ("cpm2_resized_map"); peaks: n device
 * pointers to (18, max_peaks + 1, 3) float32 joint candidates ("cpm2_joints": row 0 of a part = [count,
 * -, -], rows 1.. = (x, y, score)).  scores_dev: n * 19 * max_peaks * max_peaks float32; entry
 * [f][k][i-1][j-1] is the mean part-affinity score sum/count of candidate pair (i, j) of limb k when
 * more than min_above of the 10 samples exceed inter_threshold, else -1 (the reference's values:
 * inter_threshold 0.05, min_above 9, max_peaks 64; :795-801). */
int st_cpm2_limb_scores(st_ctx* ctx, const float* const* heatmaps_dev, const float* const* peaks_dev, int n,
                        int net_h, int net_w, int max_peaks, float inter_threshold, int min_above,
                        float* scores_dev);

/* Convolution stack of the pose network (what the Caffe forward pass behind the reference's CPM2 op
 * computes, scannertools_caffe_cpp/cpm2_kernel.cpp:8-52 / caffe_kernel.cpp; layer list: profiles/NOTES.md, Part II section 9).
 * float32 throughout, as in the reference.  Activations are NHWC float32 device arrays whose channel count
 * (x_stride / y_stride floats per pixel) is a multiple of 4; a call reads cin channels from channel
 * x_offset on and writes cout channels from channel y_offset on, so concatenations need no copy.
 * st_conv2d_nhwc_f32: stride-1 "same" convolution (odd square kernel <= 7) + bias (+ ReLU).  cin must be a
 * multiple of 16 (pad channels zero).  w_dev: [cout_pad][kh][kw][cin] with cout_pad a multiple of 64 >= cout
 * (extra rows zero), bias_dev: [cout_pad]. */
int st_conv2d_nhwc_f32(st_ctx* ctx, const float* x_dev, int n, int h, int w, int cin, int x_stride, int x_offset,
                       const float* w_dev, const float* bias_dev, int kh, int kw, int cout, int cout_pad, int relu,
                       float* y_dev, int y_stride, int y_offset);
/* The same convolution with the weights ALSO in the spatial-tile kernel's operand order (wt_dev: st_conv_f32_tile_bytes(...)
 * bytes filled once by st_conv_pack_weights_f32_tile from the same float32 tensor; the size is 0 and wt_dev may be null for
 * layers the tile kernel does not take -- anything but 3x3 / 7x7 with cout_pad a multiple of 128).  The library picks the
 * kernel per call (tile shape by map size, tile or per-tap kernel by launch size; ST_CONV_TILE=0 / 1 at st_ctx_create forces
 * the per-tap / the tile kernel).  Both accumulate every output as the same k-ordered float32 fmaf chain (16-channel slices
 * outer, taps inner, channels ascending), so the choice -- and with it the batch size -- never changes a bit.
 * st_conv2d_nhwc_f32 = this with wt_dev null. */
long long st_conv_f32_tile_bytes(int cout_pad, int kh, int kw, int cin);
int st_conv_pack_weights_f32_tile(st_ctx* ctx, const float* w_dev, int cout_pad, int kh, int kw, int cin, void* out_dev);
int st_conv2d_nhwc_f32_tiled(st_ctx* ctx, const float* x_dev, int n, int h, int w, int cin, int x_stride, int x_offset,
                             const float* w_dev, const void* wt_dev, const float* bias_dev, int kh, int kw, int cout, int cout_pad,
                             int relu, float* y_dev, int y_stride, int y_offset);
/* TWO convolutions of the same geometry (n, h, w, cin, kernel, cout_pad, relu) on different operands -- the two branches of a
 * stage of the pose network -- in one launch where the spatial-tile kernel runs (at the reference's five frames per call a
 * 7x7 layer alone leaves a third of the CUs idle), else one after the other.  The result is exactly what the two single
 * calls give.  w: the float32 tensor (f32) / the packed buffer (bf16x3); w_tile: f32 tile-order copy or null (bf16x3: unused). */
typedef struct st_conv_operands {
  const float* x; int x_stride, x_offset;
  const void* w; const void* w_tile;
  const float* bias; int cout;
  float* y; int y_stride, y_offset;
} st_conv_operands;
int st_conv2d_nhwc_f32_pair(st_ctx* ctx, int n, int h, int w, int cin, int kh, int kw, int cout_pad, int relu,
                            const st_conv_operands* a, const st_conv_operands* b);
int st_conv2d_nhwc_bf16x3_pair(st_ctx* ctx, int n, int h, int w, int cin, int kh, int kw, int cout_pad, int relu,
                               const st_conv_operands* a, const st_conv_operands* b);
/* 2x2 max pooling, stride 2: (n, h, w, c) -> (n, h/2, w/2, c), c a multiple of 4. */
/* The same convolution on the bf16 matrix pipe at float32-grade accuracy ("bf16x3": every operand split into three
 * bf16 terms, the six significant products accumulated in float32; 2.67 x the float32 matrix rate on CDNA4, results
 * within one float32 rounding per product of st_conv2d_nhwc_f32's, not bit-identical to it).  Opt-in replacement for
 * the same Caffe layers (cpm2_kernel.cpp:8-52).  Activations and outputs as above; w3_dev: the layer's weights
 * rearranged ONCE by st_conv_pack_weights_bf16x3 from the [cout_pad][kh][kw][cin] float32 tensor into a 16-byte
 * aligned buffer of st_conv_bf16x3_packed_bytes(...) bytes (6 bytes per weight; twice that for 3x3 / 7x7 layers with
 * cout_pad a multiple of 128, which also get the operand-order copy the spatial-tile kernel reads -- the library picks
 * the kernel per call; both kernels add the same products in the same order, so the result does not depend on the choice). */
long long st_conv_bf16x3_packed_bytes(int cout_pad, int kh, int kw, int cin);
/* out_bytes: the size of the buffer behind out_dev; less than st_conv_bf16x3_packed_bytes(...) is ST_ERR_INVALID (nothing is
 * written).  st_conv_pack_weights_bf16x3 is the same call without the check: the caller vouches for the size. */
int st_conv_pack_weights_bf16x3_n(st_ctx* ctx, const float* w_dev, int cout_pad, int kh, int kw, int cin, void* out_dev,
                                  size_t out_bytes);
int st_conv_pack_weights_bf16x3(st_ctx* ctx, const float* w_dev, int cout_pad, int kh, int kw, int cin, void* out_dev);
int st_conv2d_nhwc_bf16x3(st_ctx* ctx, const float* x_dev, int n, int h, int w, int cin, int x_stride, int x_offset,
                          const void* w3_dev, const float* bias_dev, int kh, int kw, int cout, int cout_pad, int relu,
                          float* y_dev, int y_stride, int y_offset);

int st_maxpool2_nhwc_f32(st_ctx* ctx, const float* x_dev, int n, int h, int w, int c, int x_stride, float* y_dev,
                         int y_stride);
/* planar (n, c, h, w) -> NHWC (n, h, w, y_stride) with zero pad channels: CPM2Input's frame as the first
 * layer's operand. */
int st_planar_to_nhwc_f32(st_ctx* ctx, const float* x_dev, int n, int c, int h, int w, float* y_dev, int y_stride);

/* The two layers between the network and CPM2Output, which the reference configures in
 * CPM2Kernel::net_config (scannertools_caffe_cpp/cpm2_kernel.cpp:13-29: the Caffe fork's ImResizeLayer "resize"
 * with start scale 1 / target = network input size, and its "nms" layer) and whose outputs are the op's columns
 * cpm2_resized_map and cpm2_joints (cpm2_kernel.cpp:46-49).  The layers' sources are not in the reference tree:
 * behaviour restated from the published fork ([EXT], unpinned; csrc/st_pose.hip says what exactly).
 * st_cpm2_resize_maps: src (n, src_h, src_w, src_stride) channel-last float32 maps -> n planar (nmaps, dst_h,
 * dst_w) frames (out_dev: host array of n device pointers); output plane c reads source channel chan_map[c]
 * (host array; NULL = identity); bicubic (Catmull-Rom), one scale. */
int st_cpm2_resize_maps(st_ctx* ctx, const float* src_dev, int n, int src_h, int src_w, int src_stride,
                        const int* chan_map, int nmaps, int dst_h, int dst_w, float* const* out_dev);
/* Several network scales merged (OpenPoseArgs.pose_num_scales / pose_scale_gap, openpose_kernel.cpp:96-112; the `num`
 * loop of the fork's resize kernel): source s = (n, src_h[s], src_w[s], src_stride) maps of which eff_h[s] x eff_w[s]
 * source pixels (float) correspond to the whole output; per output pixel the bicubic interpolants of the scales are
 * summed in order and divided by `scales` (<= 8).  One scale with eff = its map size is st_cpm2_resize_maps, bit for bit. */
int st_cpm2_resize_merge_maps(st_ctx* ctx, const float* const* src_dev, const int* src_h, const int* src_w, const float* eff_h,
                              const float* eff_w, int scales, int n, int src_stride, const int* chan_map, int nmaps, int dst_h,
                              int dst_w, float* const* out_dev);
/* st_cpm2_nms: peaks of the first `parts` planes of n (>= parts, h, w) maps: strict 8-neighbour maxima above
 * `threshold`, interior pixels only, raster order, at most max_peaks per part.  joints_dev: n device pointers to
 * (parts, max_peaks + 1, 3) float32: row 0 = [count, 0, 0], row i = (x, y, score) -- the layout
 * cpm2_output_kernel_cpu.cpp:481-499 reads. */
int st_cpm2_nms(st_ctx* ctx, const float* const* maps_dev, int n, int h, int w, int parts, int max_peaks, float threshold,
                float* const* joints_dev);

#ifdef __cplusplus
}
#endif
#endif /* SCANNERTOOLS_HIP_H_ */
