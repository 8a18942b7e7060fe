"""ctypes binding of libscannertools_hip.so (the C ABI declared in include/scannertools_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C scannertools_amd/csrc``
and lives at ``scannertools_amd/lib/libscannertools_hip.so``.  There is no fallback: if the
library is missing, :func:`lib` raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ST_HIP_LIB") or os.path.join(_HERE, "lib", "libscannertools_hip.so")
CSRC = os.path.join(_HERE, "csrc")

ST_OK, ST_ERR_INVALID, ST_ERR_HIP, ST_ERR_OOM, ST_ERR_UNSUPPORTED = range(5)
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4  # cv::InterpolationFlags values the Resize op implements
# cv::ColorConversionCodes values the ConvertColor op implements
COLOR_CODES = {"COLOR_BGR2RGB": 4, "COLOR_RGB2BGR": 4, "COLOR_BGR2GRAY": 6, "COLOR_RGB2GRAY": 7,
               "COLOR_GRAY2BGR": 8, "COLOR_GRAY2RGB": 8, "COLOR_BGR2YCrCb": 36, "COLOR_RGB2YCrCb": 37,
               "COLOR_YCrCb2BGR": 38, "COLOR_YCrCb2RGB": 39, "COLOR_BGR2HSV": 40, "COLOR_RGB2HSV": 41,
               "COLOR_HSV2BGR": 54, "COLOR_HSV2RGB": 55, "COLOR_BGR2HSV_FULL": 66, "COLOR_RGB2HSV_FULL": 67,
               "COLOR_HSV2BGR_FULL": 70, "COLOR_HSV2RGB_FULL": 71, "COLOR_BGR2YUV": 82, "COLOR_RGB2YUV": 83,
               "COLOR_BGR2HLS": 52, "COLOR_RGB2HLS": 53, "COLOR_HLS2BGR": 60, "COLOR_HLS2RGB": 61,
               "COLOR_BGR2HLS_FULL": 68, "COLOR_RGB2HLS_FULL": 69, "COLOR_HLS2BGR_FULL": 72, "COLOR_HLS2RGB_FULL": 73,
               "COLOR_YUV2BGR": 84, "COLOR_YUV2RGB": 85,
               "COLOR_BGR2XYZ": 32, "COLOR_RGB2XYZ": 33, "COLOR_XYZ2BGR": 34, "COLOR_XYZ2RGB": 35,
               # YUV 4:2:0 sources ((3H/2, W, 1) frames) and packed 4:2:2 sources ((H, W, 2) frames) -> RGB / BGR / RGBA / BGRA / gray
               "COLOR_YUV2RGB_NV12": 90, "COLOR_YUV2BGR_NV12": 91, "COLOR_YUV2RGB_NV21": 92, "COLOR_YUV2BGR_NV21": 93,
               "COLOR_YUV420sp2RGB": 92, "COLOR_YUV420sp2BGR": 93, "COLOR_YUV2RGBA_NV12": 94, "COLOR_YUV2BGRA_NV12": 95,
               "COLOR_YUV2RGBA_NV21": 96, "COLOR_YUV2BGRA_NV21": 97, "COLOR_YUV420sp2RGBA": 96, "COLOR_YUV420sp2BGRA": 97,
               "COLOR_YUV2RGB_YV12": 98, "COLOR_YUV2BGR_YV12": 99, "COLOR_YUV2RGB_IYUV": 100, "COLOR_YUV2BGR_IYUV": 101,
               "COLOR_YUV2RGB_I420": 100, "COLOR_YUV2BGR_I420": 101, "COLOR_YUV420p2RGB": 98, "COLOR_YUV420p2BGR": 99,
               "COLOR_YUV2RGBA_YV12": 102, "COLOR_YUV2BGRA_YV12": 103, "COLOR_YUV2RGBA_IYUV": 104, "COLOR_YUV2BGRA_IYUV": 105,
               "COLOR_YUV2RGBA_I420": 104, "COLOR_YUV2BGRA_I420": 105, "COLOR_YUV420p2RGBA": 102, "COLOR_YUV420p2BGRA": 103,
               "COLOR_YUV2GRAY_420": 106, "COLOR_YUV2GRAY_NV21": 106, "COLOR_YUV2GRAY_NV12": 106, "COLOR_YUV2GRAY_YV12": 106,
               "COLOR_YUV2GRAY_IYUV": 106, "COLOR_YUV2GRAY_I420": 106, "COLOR_YUV420sp2GRAY": 106, "COLOR_YUV420p2GRAY": 106,
               "COLOR_YUV2RGB_UYVY": 107, "COLOR_YUV2BGR_UYVY": 108, "COLOR_YUV2RGB_Y422": 107, "COLOR_YUV2BGR_Y422": 108,
               "COLOR_YUV2RGB_UYNV": 107, "COLOR_YUV2BGR_UYNV": 108, "COLOR_YUV2RGBA_UYVY": 111, "COLOR_YUV2BGRA_UYVY": 112,
               "COLOR_YUV2RGBA_Y422": 111, "COLOR_YUV2BGRA_Y422": 112, "COLOR_YUV2RGBA_UYNV": 111, "COLOR_YUV2BGRA_UYNV": 112,
               "COLOR_YUV2RGB_YUY2": 115, "COLOR_YUV2BGR_YUY2": 116, "COLOR_YUV2RGB_YVYU": 117, "COLOR_YUV2BGR_YVYU": 118,
               "COLOR_YUV2RGB_YUYV": 115, "COLOR_YUV2BGR_YUYV": 116, "COLOR_YUV2RGB_YUNV": 115, "COLOR_YUV2BGR_YUNV": 116,
               "COLOR_YUV2RGBA_YUY2": 119, "COLOR_YUV2BGRA_YUY2": 120, "COLOR_YUV2RGBA_YVYU": 121, "COLOR_YUV2BGRA_YVYU": 122,
               "COLOR_YUV2RGBA_YUYV": 119, "COLOR_YUV2BGRA_YUYV": 120, "COLOR_YUV2RGBA_YUNV": 119, "COLOR_YUV2BGRA_YUNV": 120,
               "COLOR_YUV2GRAY_UYVY": 123, "COLOR_YUV2GRAY_YUY2": 124, "COLOR_YUV2GRAY_Y422": 123, "COLOR_YUV2GRAY_UYNV": 123,
               "COLOR_YUV2GRAY_YVYU": 124, "COLOR_YUV2GRAY_YUYV": 124, "COLOR_YUV2GRAY_YUNV": 124,
               # channel layout family (alpha channel, 16-bit packed pixels)
               "COLOR_BGR2BGRA": 0, "COLOR_RGB2RGBA": 0, "COLOR_BGRA2BGR": 1, "COLOR_RGBA2RGB": 1, "COLOR_BGR2RGBA": 2,
               "COLOR_RGB2BGRA": 2, "COLOR_RGBA2BGR": 3, "COLOR_BGRA2RGB": 3, "COLOR_BGRA2RGBA": 5, "COLOR_RGBA2BGRA": 5,
               "COLOR_GRAY2BGRA": 9, "COLOR_GRAY2RGBA": 9, "COLOR_BGRA2GRAY": 10, "COLOR_RGBA2GRAY": 11,
               "COLOR_BGR2BGR565": 12, "COLOR_RGB2BGR565": 13, "COLOR_BGR5652BGR": 14, "COLOR_BGR5652RGB": 15,
               "COLOR_BGRA2BGR565": 16, "COLOR_RGBA2BGR565": 17, "COLOR_BGR5652BGRA": 18, "COLOR_BGR5652RGBA": 19,
               "COLOR_GRAY2BGR565": 20, "COLOR_BGR5652GRAY": 21, "COLOR_BGR2BGR555": 22, "COLOR_RGB2BGR555": 23,
               "COLOR_BGR5552BGR": 24, "COLOR_BGR5552RGB": 25, "COLOR_BGRA2BGR555": 26, "COLOR_RGBA2BGR555": 27,
               "COLOR_BGR5552BGRA": 28, "COLOR_BGR5552RGBA": 29, "COLOR_GRAY2BGR555": 30, "COLOR_BGR5552GRAY": 31}
K_HIST, K_GRAY, K_PYR, K_POLYEXP, K_UPDATE_MATRICES, K_BLUR_UPDATE, K_FLOW_HIST, K_DRAW_FLOW, K_BLUR_OP, K_RESIZE, K_CVT_COLOR, K_CPM2_INPUT, K_CPM2_LIMBS, K_CONV, K_CPM2_RESIZE, K_CPM2_NMS, K_COUNT = range(17)
KERNEL_NAMES = ["hist", "gray", "pyr", "polyexp", "update_matrices", "blur_update", "flow_hist", "draw_flow", "blur_op", "resize", "cvt_color", "cpm2_input", "cpm2_limbs", "conv", "cpm2_resize", "cpm2_nms"]
assert len(KERNEL_NAMES) == K_COUNT


class StError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("scannertools_hip status %d: %s" % (status, msg))
        self.status = status


class ConvOperands(ctypes.Structure):
    """st_conv_operands (include/scannertools_hip.h): one of the two convolutions of a paired launch."""
    _fields_ = [("x", ctypes.c_void_p), ("x_stride", ctypes.c_int), ("x_offset", ctypes.c_int),
                ("w", ctypes.c_void_p), ("w_tile", ctypes.c_void_p),
                ("bias", ctypes.c_void_p), ("cout", ctypes.c_int),
                ("y", ctypes.c_void_p), ("y_stride", ctypes.c_int), ("y_offset", ctypes.c_int)]


class FbParams(ctypes.Structure):
    """``st_fb_params``: arguments of cv::FarnebackOpticalFlow::create as the reference passes
    them (scannertools_cpp/imgproc/optical_flow_kernel_cpu.cpp:15-16) + the gray table width."""
    _fields_ = [("num_levels", ctypes.c_int), ("pyr_scale", ctypes.c_double),
                ("fast_pyramids", ctypes.c_int), ("win_size", ctypes.c_int),
                ("num_iters", ctypes.c_int), ("poly_n", ctypes.c_int),
                ("poly_sigma", ctypes.c_double), ("flags", ctypes.c_int),
                ("gray_bits", ctypes.c_int)]


# symbol -> (restype, argtypes); also the list the CPU test-suite checks against the header
_c = ctypes
_vp, _i, _sz, _d = _c.c_void_p, _c.c_int, _c.c_size_t, _c.c_double
SIGNATURES = {
    "st_abi_version": (_i, []),
    "st_build_info": (_c.c_char_p, []),
    "st_status_string": (_c.c_char_p, [_i]),
    "st_device_count": (_i, [_c.POINTER(_i)]),
    "st_ctx_create": (_i, [_i, _c.POINTER(_vp)]),
    "st_ctx_destroy": (_i, [_vp]),
    "st_ctx_set_stream": (_i, [_vp, _vp]),
    "st_ctx_reset_stream": (_i, [_vp]),
    "st_ctx_sync": (_i, [_vp]),
    "st_ctx_flow_concurrent": (_i, [_vp]),
    "st_ctx_set_workspace_limit": (_i, [_vp, _sz]),
    "st_ctx_release_workspace": (_i, [_vp]),
    "st_ctx_last_error": (_c.c_char_p, [_vp]),
    "st_ctx_timing_enable": (_i, [_vp, _c.c_uint]),
    "st_ctx_timing_reset": (_i, [_vp]),
    "st_ctx_timing_read": (_i, [_vp, _i, _c.POINTER(_i), _c.POINTER(_d)]),
    "st_hist_u8c3_batch": (_i, [_vp, _c.POINTER(_vp), _i, _i, _i, _i, _vp]),
    "st_hist_u8c3_strided": (_i, [_vp, _vp, _sz, _i, _i, _i, _i, _vp]),
    "st_shot_boundaries": (_i, [_vp, _vp, _i, _i, _i, _d, _vp, _vp]),
    "st_fb_params_default": (None, [_c.POINTER(FbParams)]),
    "st_farneback_pairs": (_i, [_vp, _c.POINTER(_vp), _i, _c.POINTER(_c.c_int32), _i, _i, _i,
                                _c.POINTER(FbParams), _c.POINTER(_vp)]),
    "st_fb_levels": (_i, [_i, _i, _c.POINTER(FbParams)]),
    "st_fb_level_geom": (_i, [_i, _i, _c.POINTER(FbParams), _i, _c.POINTER(_i), _c.POINTER(_i),
                              _c.POINTER(_d), _c.POINTER(_i)]),
    "st_gray_u8": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "st_fb_pyr_image": (_i, [_vp, _vp, _i, _i, _c.POINTER(FbParams), _i, _vp]),
    "st_fb_polyexp": (_i, [_vp, _vp, _i, _i, _i, _d, _vp]),
    "st_fb_update_matrices": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _d, _i, _i, _vp]),
    "st_fb_update_flow_blur": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "st_fb_flow_iteration": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _d, _i, _i, _i, _vp]),
    "st_flow_hist_batch": (_i, [_vp, _c.POINTER(_vp), _i, _i, _i, _vp]),
    "st_flow_hist_strided": (_i, [_vp, _vp, _sz, _i, _i, _i, _vp]),
    "st_draw_flow_batch": (_i, [_vp, _c.POINTER(_vp), _c.POINTER(_vp), _i, _i, _i, _c.POINTER(_vp)]),
    "st_box_blur_u8c3_batch": (_i, [_vp, _c.POINTER(_vp), _i, _i, _i, _i, _c.POINTER(_vp)]),
    "st_resize_u8_batch": (_i, [_vp, _c.POINTER(_vp), _i, _i, _i, _i, _i, _i, _i, _c.POINTER(_vp)]),
    "st_cvt_color_out_channels": (_i, [_i, _i]),
    "st_cvt_color_out_shape": (_i, [_i, _i, _i, _i, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i)]),
    "st_cvt_color_u8_batch": (_i, [_vp, _c.POINTER(_vp), _i, _i, _i, _i, _i, _i, _c.POINTER(_vp)]),
    "st_cpm2_geometry": (_i, [_i, _i, _c.c_float, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i)]),
    "st_cpm2_scale_for_height": (_i, [_i, _i, _c.POINTER(_c.c_float)]),
    "st_cpm2_input_batch": (_i, [_vp, _c.POINTER(_vp), _i, _i, _i, _c.c_float, _c.POINTER(_vp)]),
    "st_cpm2_limb_scores": (_i, [_vp, _c.POINTER(_vp), _c.POINTER(_vp), _i, _i, _i, _i, _c.c_float, _i, _vp]),
    "st_conv2d_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i]),
    "st_conv_bf16x3_packed_bytes": (ctypes.c_longlong, [_i, _i, _i, _i]),
    "st_conv_f32_tile_bytes": (ctypes.c_longlong, [_i, _i, _i, _i]),
    "st_conv_pack_weights_f32_tile": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "st_conv2d_nhwc_f32_tiled": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i]),
    "st_conv_pack_weights_bf16x3": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "st_conv_pack_weights_bf16x3_n": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, ctypes.c_size_t]),
    "st_conv2d_nhwc_f32_pair": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _c.POINTER(ConvOperands), _c.POINTER(ConvOperands)]),
    "st_conv2d_nhwc_bf16x3_pair": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _c.POINTER(ConvOperands), _c.POINTER(ConvOperands)]),
    "st_conv2d_nhwc_bf16x3": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i]),
    "st_maxpool2_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i]),
    "st_planar_to_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i]),
    "st_cpm2_resize_maps": (_i, [_vp, _vp, _i, _i, _i, _i, _c.POINTER(_i), _i, _i, _i, _c.POINTER(_vp)]),
    "st_cpm2_resize_merge_maps": (_i, [_vp, _c.POINTER(_vp), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_c.c_float), _c.POINTER(_c.c_float), _i, _i,
                                       _i, _c.POINTER(_i), _i, _i, _i, _c.POINTER(_vp)]),
    "st_cpm2_nms": (_i, [_vp, _c.POINTER(_vp), _i, _i, _i, _i, _i, _c.c_float, _c.POINTER(_vp)]),
}

_LIB = None


def build(verbose=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=out)
    return LIB_PATH


def source_hash():
    """First 16 hex digits of the sha256 over the library's sources as the Makefile computes it (`make srchash`): what
    st_build_info() of a library built from THIS tree reports."""
    import hashlib
    srcs = ["st_context.hip", "st_hist.hip", "st_farneback.hip", "st_flowvis.hip", "st_imgproc.hip", "st_pose.hip", "st_conv.hip",
            "st_conv_tile_bf16x3.hip", "st_conv_tile_f32.hip", "st_internal.h", "st_conv_tile.h",
            os.path.join("..", "..", "include", "scannertools_hip.h"), "Makefile"]
    hsh = hashlib.sha256()
    for f in srcs:
        hsh.update(open(os.path.join(CSRC, f), "rb").read())
    return hsh.hexdigest()[:16]


def build_info():
    """{'src': ..., 'host': ..., 'at': ...} of the loaded library (st_build_info)."""
    return dict(kv.split("=", 1) for kv in lib().st_build_info().decode().split(" "))


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libscannertools_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C scannertools_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        import torch  # noqa: F401  -- load torch's HIP runtime first so both share one libamdhip64
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.st_abi_version() != 1:
            raise RuntimeError("libscannertools_hip.so ABI version mismatch")
        _LIB = L
    return _LIB


def default_params(**overrides):
    p = FbParams()
    lib().st_fb_params_default(ctypes.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise TypeError("unknown Farneback parameter %r" % k)
        setattr(p, k, v)
    return p
