"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package, and only as the checker / CPU baseline.  Nothing under
``scannertools_amd/`` imports it: the product path fails loudly when its HIP library is
missing instead of falling back to this code.

``oracle.c`` restates the OpenCV algorithms the reference ops call (see its header for the
reference call sites and the parity status: Histogram pinned by definition, DrawFlow pinned by
golden vectors from the reference's vis.py, Blur pinned by the reference source itself;
OpticalFlow, FlowHistogram, Resize and ConvertColor PARITY UNPINNED against real OpenCV
output); ``shot_boundaries`` restates
``/root/reference/scannertools/scannertools/shot_detection.py:11-28`` and is pinned by the
fixtures under ``tests/golden/`` that were produced by importing that file.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FbParams(ctypes.Structure):
    """Mirror of ``orc_fb_params``; defaults = the reference's
    ``FarnebackOpticalFlow::create(3, 0.5, false, 15, 3, 5, 1.2, 0)``
    (scannertools_cpp/imgproc/optical_flow_kernel_cpu.cpp:16)."""
    _fields_ = [("num_levels", ctypes.c_int), ("pyr_scale", ctypes.c_double),
                ("fast_pyramids", ctypes.c_int), ("win_size", ctypes.c_int),
                ("num_iters", ctypes.c_int), ("poly_n", ctypes.c_int),
                ("poly_sigma", ctypes.c_double), ("flags", ctypes.c_int),
                ("gray_bits", ctypes.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_fb_levels.restype = ctypes.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def default_params(**kw):
    p = FbParams()
    lib().orc_fb_params_default(ctypes.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def hist_u8c3(frame, bins=16):
    """(h,w,3) uint8 -> (3,bins) int32 (A0/A1)."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w, c = frame.shape
    assert c == 3
    out = np.empty((3, bins), np.int32)
    lib().orc_hist_u8c3(_p(frame), h, w, bins, _p(out))
    return out


def gray_u8(frame, bits=15):
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w, _ = frame.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_gray_u8(_p(frame), h, w, bits, _p(out))
    return out


def gaussian_kernel(n, sigma):
    k = np.empty(n, np.float32)
    lib().orc_gaussian_kernel(n, ctypes.c_double(sigma), _p(k))
    return k


def gaussian_blur(img, ks, sigma):
    img = np.ascontiguousarray(img, dtype=np.float32)
    out = np.empty_like(img)
    lib().orc_gaussian_blur_f32(_p(img), img.shape[0], img.shape[1], ks, ctypes.c_double(sigma), _p(out))
    return out


def resize_linear(img, dh, dw):
    img = np.ascontiguousarray(img, dtype=np.float32)
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.float32)
    lib().orc_resize_linear_f32(_p(img), img.shape[0], img.shape[1], cn, _p(out), dh, dw)
    return out


def poly_prepare(n=5, sigma=1.2):
    buf = np.zeros((3, 2 * n + 1), np.float32)
    ig = np.zeros(4, np.float64)
    base = buf.ctypes.data
    sz = 4
    g = ctypes.c_void_p(base + n * sz)
    xg = ctypes.c_void_p(base + (2 * n + 1 + n) * sz)
    xxg = ctypes.c_void_p(base + (2 * (2 * n + 1) + n) * sz)
    lib().orc_poly_prepare(n, ctypes.c_double(sigma), g, xg, xxg, _p(ig))
    return buf, ig


def polyexp(I, n=5, sigma=1.2):
    I = np.ascontiguousarray(I, dtype=np.float32)
    h, w = I.shape
    R = np.empty((h, w, 5), np.float32)
    lib().orc_polyexp(_p(I), h, w, n, ctypes.c_double(sigma), _p(R))
    return R


def update_matrices(R0, R1, flow):
    R0 = np.ascontiguousarray(R0, np.float32)
    R1 = np.ascontiguousarray(R1, np.float32)
    flow = np.ascontiguousarray(flow, np.float32)
    h, w, _ = R0.shape
    M = np.empty((h, w, 5), np.float32)
    lib().orc_update_matrices(_p(R0), _p(R1), _p(flow), _p(M), h, w, 0, h)
    return M


def update_flow_blur(R0, R1, M, block_size=15, update=True):
    """Returns (flow, M') -- M' is M updated in place when ``update``."""
    R0 = np.ascontiguousarray(R0, np.float32)
    R1 = np.ascontiguousarray(R1, np.float32)
    M = np.array(M, np.float32, copy=True, order="C")
    h, w, _ = R0.shape
    flow = np.zeros((h, w, 2), np.float32)
    lib().orc_update_flow_blur(_p(R0), _p(R1), _p(flow), _p(M), h, w, block_size, int(bool(update)))
    return flow, M


def fb_levels(h, w, params=None):
    p = params or default_params()
    return lib().orc_fb_levels(h, w, ctypes.byref(p))


def fb_level_geom(h, w, k, params=None):
    p = params or default_params()
    lh, lw, ks = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    sg = ctypes.c_double()
    lib().orc_fb_level_geom(h, w, ctypes.byref(p), k, ctypes.byref(lh), ctypes.byref(lw),
                            ctypes.byref(sg), ctypes.byref(ks))
    return lh.value, lw.value, sg.value, ks.value


def fb_pyr_image(gray, k, params=None):
    p = params or default_params()
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    lh, lw, _, _ = fb_level_geom(h, w, k, p)
    out = np.empty((lh, lw), np.float32)
    lib().orc_fb_pyr_image(_p(gray), h, w, ctypes.byref(p), k, _p(out))
    return out


def farneback(prev_gray, next_gray, params=None):
    p = params or default_params()
    a = np.ascontiguousarray(prev_gray, np.uint8)
    b = np.ascontiguousarray(next_gray, np.uint8)
    h, w = a.shape
    flow = np.empty((h, w, 2), np.float32)
    lib().orc_farneback(_p(a), _p(b), h, w, ctypes.byref(p), _p(flow))
    return flow


def optical_flow_rgb(frame0, frame1, params=None):
    """The OpticalFlow op on two RGB frames: stencil element 0 -> element 1."""
    p = params or default_params()
    a = np.ascontiguousarray(frame0, np.uint8)
    b = np.ascontiguousarray(frame1, np.uint8)
    h, w, _ = a.shape
    flow = np.empty((h, w, 2), np.float32)
    lib().orc_optical_flow_rgb(_p(a), _p(b), h, w, ctypes.byref(p), _p(flow))
    return flow


WINDOW_SIZE = 500  # shot_detection.py:7


def cart_to_polar_deg(flow):
    """(h,w,2) float32 -> (mag, deg) float32, cv::cartToPolar(angleInDegrees=true) restated."""
    flow = np.ascontiguousarray(flow, dtype=np.float32)
    h, w, _ = flow.shape
    mag = np.empty((h, w), np.float32)
    deg = np.empty((h, w), np.float32)
    lib().orc_cart_to_polar_deg(_p(flow), ctypes.c_size_t(h * w), _p(mag), _p(deg))
    return mag, deg


def flow_hist(flow):
    """(h,w,2) float32 -> (2,64) int32: FlowHistogram (old/cpp_ops/flow_histogram_kernel_cpu.cpp:26-57)."""
    flow = np.ascontiguousarray(flow, dtype=np.float32)
    h, w, c = flow.shape
    assert c == 2
    out = np.empty((2, 64), np.int32)
    lib().orc_flow_hist(_p(flow), h, w, _p(out))
    return out


def draw_flow(frame, flow):
    """(h,w,3) uint8, (h,w,2) float32 -> (h,2w,3) uint8: DrawFlow (scannertools/vis.py:8-12)."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    flow = np.ascontiguousarray(flow, dtype=np.float32)
    h, w, _ = frame.shape
    assert flow.shape == (h, w, 2)
    out = np.empty((h, 2 * w, 3), np.uint8)
    lib().orc_draw_flow(_p(frame), _p(flow), h, w, _p(out))
    return out


def box_blur(frame, kernel_size):
    """(h,w,3) uint8 -> (h,w,3) uint8: the Blur op (blur_kernel_cpu.cpp:36-81); border pixels 0."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w, c = frame.shape
    assert c == 3
    out = np.empty_like(frame)
    lib().orc_box_blur_u8c3(_p(frame), h, w, int(kernel_size), _p(out))
    return out


INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4


def resize_target(frame_w, frame_h, width=0, height=0, min=False, preserve_aspect=False):
    """Target (width, height) as ResizeKernel::execute derives it (resize_kernel.cpp:44-62)."""
    tw, th = ctypes.c_int(), ctypes.c_int()
    lib().orc_resize_target(frame_w, frame_h, width, height, int(min), int(preserve_aspect), ctypes.byref(tw), ctypes.byref(th))
    return tw.value, th.value


def resize_u8(frame, width, height, interpolation=INTER_LINEAR):
    """(h,w,c) uint8 -> (height,width,c) uint8: cv::resize restated for 8-bit frames."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w, c = frame.shape
    out = np.empty((height, width, c), np.uint8)
    if lib().orc_resize_u8(_p(frame), h, w, c, _p(out), height, width, interpolation):
        raise ValueError("unsupported interpolation %r" % interpolation)
    return out


COLOR_BGR2RGB, COLOR_RGB2BGR, COLOR_BGR2GRAY, COLOR_RGB2GRAY, COLOR_GRAY2BGR, COLOR_GRAY2RGB, COLOR_BGR2HSV = 4, 4, 6, 7, 8, 8, 40
COLOR_BGR2YCrCb, COLOR_RGB2YCrCb, COLOR_YCrCb2BGR, COLOR_YCrCb2RGB = 36, 37, 38, 39
COLOR_RGB2HSV, COLOR_HSV2BGR, COLOR_HSV2RGB = 41, 54, 55
COLOR_BGR2HSV_FULL, COLOR_RGB2HSV_FULL, COLOR_HSV2BGR_FULL, COLOR_HSV2RGB_FULL = 66, 67, 70, 71
COLOR_BGR2HLS, COLOR_RGB2HLS, COLOR_HLS2BGR, COLOR_HLS2RGB = 52, 53, 60, 61
COLOR_BGR2HLS_FULL, COLOR_RGB2HLS_FULL, COLOR_HLS2BGR_FULL, COLOR_HLS2RGB_FULL = 68, 69, 72, 73
COLOR_BGR2YUV, COLOR_RGB2YUV, COLOR_YUV2BGR, COLOR_YUV2RGB = 82, 83, 84, 85
COLOR_BGR2XYZ, COLOR_RGB2XYZ, COLOR_XYZ2BGR, COLOR_XYZ2RGB = 32, 33, 34, 35


def cvt_color(frame, code, gray_bits=15):
    """(h,w,c) uint8 -> (h',w,c') uint8: cv::cvtColor restated for the codes above (h' = 2h/3 for the YUV 4:2:0 sources)."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w, c = frame.shape
    oh, ow, oc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    if lib().orc_cvt_color_out_shape(int(code), h, w, c, ctypes.byref(oh), ctypes.byref(ow), ctypes.byref(oc)) != 0:
        raise ValueError("unsupported conversion %r for a %dx%d frame of %d channel(s)" % (code, w, h, c))
    out = np.empty((oh.value, ow.value, oc.value), np.uint8)
    assert lib().orc_cvt_color_u8(_p(frame), h, w, c, int(code), gray_bits, _p(out)) == 0
    return out


def cpm2_geometry(h, w, scale):
    """(resize_h, resize_w, net_h, net_w) as cpm2_input_kernel_gpu.cpp:44-55 derives them."""
    v = [ctypes.c_int() for _ in range(4)]
    if lib().orc_cpm2_geometry(h, w, ctypes.c_float(scale), *[ctypes.byref(x) for x in v]):
        raise ValueError("empty network input")
    return tuple(x.value for x in v)


def cpm2_input(frame, scale):
    """(h,w,3) uint8 RGB -> (3,net_h,net_w) float32: the CPM2Input op."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w, _ = frame.shape
    _, _, nh, nw = cpm2_geometry(h, w, scale)
    out = np.empty((3, nh, nw), np.float32)
    assert lib().orc_cpm2_input(_p(frame), h, w, ctypes.c_float(scale), _p(out)) == 0
    return out


# COCO_18 tables of the CPM2Output op (cpm2_output_kernel_cpu.cpp:84-88)
CPM2_LIMB_SEQ = [1, 2, 1, 5, 2, 3, 3, 4, 5, 6, 6, 7, 1, 8, 8, 9, 9, 10, 1, 11, 11, 12, 12, 13, 1, 0, 0, 14, 14, 16, 0, 15, 15,
                 17, 2, 16, 5, 17]
CPM2_MAP_IDX = [31, 32, 39, 40, 33, 34, 35, 36, 41, 42, 43, 44, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 47, 48, 49,
                50, 53, 54, 51, 52, 55, 56, 37, 38, 45, 46]


def _rm_taps(pos, length):
    """Neighbour indices of the Caffe fork's bicubic resize for float32 positions `pos` ([EXT], see cpm2_resize_maps)."""
    c = (pos.astype(np.float64) + 1e-5).astype(np.int64)   # int(x + 1e-5): truncation towards zero
    c = np.maximum(c, 0)
    n0 = np.where(c - 1 < 0, c, c - 1)
    n2 = np.where(c + 1 >= length, length - 1, c + 1)
    n3 = np.where(n2 + 1 >= length, length - 1, n2 + 1)
    return n0, c, n2, n3


def _rm_cubic(v0, v1, v2, v3, d):
    f = np.float32
    return ((f(-0.5) * v0 + f(1.5) * v1 - f(1.5) * v2 + f(0.5) * v3) * d * d * d + (v0 - f(2.5) * v1 + f(2.0) * v2 - f(0.5) * v3) * d * d
            + (f(-0.5) * v0 + f(0.5) * v2) * d + v1)


def cpm2_resize_maps(maps, dst_h, dst_w, eff=None):
    """The `resize` layer of the Caffe fork behind the reference's CPM2 op (ImResizeLayer with start scale 1 and one
    scale, configured at scannertools_caffe_cpp/cpm2_kernel.cpp:16-23).  The layer's source is not in the reference
    tree: restated from the published caffe_rtpose kernel ([EXT], PARITY UNPINNED).  maps: (C, h, w) float32 ->
    (C, dst_h, dst_w) float32, every operation in float32 in the layer's order (the 1e-5 nudge and the offset's
    `- 0.5` in double, as the C expression evaluates them).  eff = (eff_h, eff_w): the source extent (float) that spans
    the whole output, when it is not the map itself (a smaller network scale, cpm2_resize_merge_maps)."""
    maps = np.ascontiguousarray(maps, dtype=np.float32)
    C, h, w = maps.shape
    f = np.float32
    eh, ew = (f(h), f(w)) if eff is None else (f(eff[0]), f(eff[1]))
    off_x = f(np.float64(f(f(dst_w) / ew) / f(2)) - 0.5)
    off_y = f(np.float64(f(f(dst_h) / eh) / f(2)) - 0.5)
    rx, ry = ew / f(dst_w), eh / f(dst_h)
    x_on = (np.arange(dst_w, dtype=np.float32) - off_x) * rx
    y_on = (np.arange(dst_h, dtype=np.float32) - off_y) * ry
    xn = _rm_taps(x_on, w)
    yn = _rm_taps(y_on, h)
    dx = (x_on - xn[1].astype(np.float32))[None, None, :]
    dy = (y_on - yn[1].astype(np.float32))[None, :, None]
    rows = []
    for i in range(4):
        r = maps[:, yn[i], :]                                  # (C, dst_h, w)
        rows.append(_rm_cubic(r[:, :, xn[0]], r[:, :, xn[1]], r[:, :, xn[2]], r[:, :, xn[3]], dx))
    return _rm_cubic(rows[0], rows[1], rows[2], rows[3], dy).astype(np.float32)


def cpm2_resize_merge_maps(maps_list, eff_sizes, dst_h, dst_w):
    """Several network scales merged ([EXT]: the `num` loop of the fork's resize kernel / OpenPose's resizeAndMerge;
    OpenPoseArgs.pose_num_scales, openpose_kernel.cpp:110-111): the scales' interpolants summed in order, divided by their
    number, in float32."""
    total = None
    for m, e in zip(maps_list, eff_sizes):
        v = cpm2_resize_maps(m, dst_h, dst_w, eff=e)
        total = v if total is None else total + v
    return (total / np.float32(len(maps_list))).astype(np.float32)


def cpm2_nms(maps, parts=18, max_peaks=64, threshold=0.05):
    """The `nms` layer of the same fork ([EXT], PARITY UNPINNED): per part plane, interior pixels above `threshold`
    that exceed their 8 neighbours, raster order, the first max_peaks kept.  Returns (parts, max_peaks + 1, 3)
    float32: row 0 = [count, 0, 0], row i = (x, y, score) -- the layout cpm2_output_kernel_cpu.cpp:481-499 reads."""
    maps = np.asarray(maps, dtype=np.float32)
    out = np.zeros((parts, max_peaks + 1, 3), np.float32)
    for p in range(parts):
        m = maps[p]
        c = m[1:-1, 1:-1]
        peak = c > np.float32(threshold)
        for dy in (0, 1, 2):
            for dx in (0, 1, 2):
                if dy == 1 and dx == 1:
                    continue
                peak &= c > m[dy:dy + c.shape[0], dx:dx + c.shape[1]]
        ys, xs = np.nonzero(peak)                              # row-major = raster order
        k = min(len(ys), max_peaks)
        out[p, 0, 0] = k
        out[p, 1:k + 1, 0] = xs[:k] + 1
        out[p, 1:k + 1, 1] = ys[:k] + 1
        out[p, 1:k + 1, 2] = c[ys[:k], xs[:k]]
    return out


def _c_round(x):
    """C round(): half away from zero."""
    import math
    x = float(x)
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


def cpm2_limb_scores(heatmap, peaks, inter_threshold=0.05, min_above=9):
    """Candidate scoring of connect_limbs_coco (cpm2_output_kernel_cpu.cpp:424-487) in float32, pair
    by pair: heatmap (57,H,W), peaks (18,max_peaks+1,3) -> (19,max_peaks,max_peaks), -1 = rejected."""
    f32 = np.float32
    heatmap = np.asarray(heatmap, f32)
    pk = np.asarray(peaks, f32)
    _, H, W = heatmap.shape
    mp = pk.shape[1] - 1
    out = np.full((19, mp, mp), -1, f32)
    thr = f32(inter_threshold)
    for k in range(19):
        map_x, map_y = heatmap[CPM2_MAP_IDX[2 * k]].ravel(), heatmap[CPM2_MAP_IDX[2 * k + 1]].ravel()
        candA, candB = pk[CPM2_LIMB_SEQ[2 * k]].ravel(), pk[CPM2_LIMB_SEQ[2 * k + 1]].ravel()
        nA, nB = int(candA[0]), int(candB[0])
        for i in range(1, nA + 1):
            for j in range(1, nB + 1):
                s_x, s_y = candA[i * 3], candA[i * 3 + 1]
                d_x, d_y = f32(candB[j * 3] - candA[i * 3]), f32(candB[j * 3 + 1] - candA[i * 3 + 1])
                norm_vec = np.sqrt(f32(f32(d_x * d_x) + f32(d_y * d_y)))
                if norm_vec < 1e-6:
                    continue
                vec_x, vec_y = f32(d_x / norm_vec), f32(d_y / norm_vec)
                total, count = f32(0), 0
                for lm in range(10):
                    my = _c_round(f32(s_y + f32(f32(f32(lm) * d_y) / f32(10))))
                    mx = _c_round(f32(s_x + f32(f32(f32(lm) * d_x) / f32(10))))
                    mx, my = min(mx, W - 1), min(my, H - 1)
                    assert mx >= 0 and my >= 0                     # CHECK_GE in the reference
                    idx = my * W + mx
                    score = f32(f32(vec_x * map_x[idx]) + f32(vec_y * map_y[idx]))
                    if score > thr:
                        total = f32(total + score)
                        count += 1
                if count > min_above:
                    out[k, i - 1, j - 1] = f32(total / f32(count))
    return out


def cpm2_connect_limbs_coco(heatmap, peaks, frame_h, frame_w, scores=None, max_people=96, min_subset_cnt=3,
                            min_subset_score=0.4, inter_threshold=0.05, min_above=9):
    """connect_limbs_coco (cpm2_output_kernel_cpu.cpp:362-689) row by row, with the reference's row-of-
    doubles bookkeeping: people as float32 (n,18,3) in original-frame coordinates.  Candidates are
    ordered with a stable sort (the reference's std::sort leaves ties unspecified)."""
    f32 = np.float32
    pk = np.asarray(peaks, f32)
    _, H, W = np.asarray(heatmap).shape
    mp = pk.shape[1] - 1
    flat = pk.ravel()
    if scores is None:
        scores = cpm2_limb_scores(heatmap, peaks, inter_threshold, min_above)
    num_parts, off = 18, 3 * (mp + 1)
    CNT, SCORE, SIZE = num_parts + 2, num_parts + 1, num_parts + 3
    subset = []
    for k in range(19):
        pa, pb = CPM2_LIMB_SEQ[2 * k], CPM2_LIMB_SEQ[2 * k + 1]
        candA, candB = pk[pa].ravel(), pk[pb].ravel()
        nA, nB = int(candA[0]), int(candB[0])
        if nA == 0 and nB == 0:
            continue
        if nA == 0 or nB == 0:
            part, cand, n = (pb, candB, nB) if nA == 0 else (pa, candA, nA)
            for i in range(1, n + 1):
                o = part * off + i * 3 + 2
                if not any(row[part] == o for row in subset):
                    row = [0.0] * SIZE
                    row[part], row[CNT], row[SCORE] = float(o), 1.0, float(cand[i * 3 + 2])
                    subset.append(row)
            continue
        temp = [(i, j, float(scores[k, i - 1, j - 1])) for i in range(1, nA + 1) for j in range(1, nB + 1)
                if scores[k, i - 1, j - 1] >= 0]
        temp.sort(key=lambda r: -r[2])
        num, usedA, usedB, conns = min(nA, nB), set(), set(), []
        for i, j, sc in temp:
            if len(conns) == num:
                break
            if i not in usedA and j not in usedB:
                conns.append((pa * off + i * 3 + 2, pb * off + j * 3 + 2, float(f32(sc))))
                usedA.add(i); usedB.add(j)
        for a, b, sc in conns:
            hits = 0
            if k != 0:
                for row in subset:
                    if row[pa] == a:
                        row[pb] = float(b)
                        hits += 1
                        row[CNT] += 1
                        row[SCORE] = row[SCORE] + float(flat[b]) + sc
            if hits == 0:
                row = [0.0] * SIZE
                row[pa], row[pb], row[CNT] = float(a), float(b), 2.0
                row[SCORE] = float(f32(flat[a] + flat[b])) + sc
                subset.append(row)
    people = []
    for row in subset:
        if row[CNT] >= min_subset_cnt and row[SCORE] / row[CNT] > float(f32(min_subset_score)):
            j3 = np.zeros((num_parts, 3), f32)
            for j in range(num_parts):
                idx = int(row[j])
                if idx:
                    j3[j, 2] = flat[idx]
                    j3[j, 1] = f32(f32(flat[idx - 1] * f32(frame_h)) / f32(H))
                    j3[j, 0] = f32(f32(flat[idx - 2] * f32(frame_w)) / f32(W))
            people.append(j3)
            if len(people) == max_people:
                break
    return np.stack(people) if people else np.zeros((0, num_parts, 3), f32)


def shot_boundaries(histograms):
    """Restatement of shot_detection.py:11-28 (A8).  ``histograms``: sequence of N items,
    each indexable as [channel][bin] (3 channels).  Returns the list of boundary indices
    (row 0 of the reference's output; rows 1.. are None there)."""
    h = np.asarray(histograms)
    n = len(h)
    if n == 0:
        return []
    hh = h.astype(np.float64)  # scipy chebyshev validates to double before max|u-v|
    diffs = np.array([np.mean([np.max(np.abs(hh[i - 1][j] - hh[i][j])) for j in range(3)])
                      for i in range(1, n)])
    diffs = np.insert(diffs, 0, 0)
    out = []
    for i in range(1, n):
        win = diffs[max(i - WINDOW_SIZE, 0):min(i + WINDOW_SIZE, n)]
        if diffs[i] - np.mean(win) > 2.5 * np.std(win):
            out.append(i)
    return out
