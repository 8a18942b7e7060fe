"""Sibling imgproc ops micro benchmark on device-resident 1080p frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scannertools_amd import _native
from scannertools_amd.hip import HipContext

ctx = HipContext(0)
n, h, w = int(os.environ.get("N", 64)), 1080, 1920
g = torch.Generator(device="cuda").manual_seed(0)
frames = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
for k in (3, 5, 7, 9, 11, 13, 15, 17):
    out = ctx.box_blur(frames, k)
    ctx.timing_enable([_native.K_BLUR_OP]); ctx.timing_reset()
    for _ in range(10):
        ctx.box_blur(frames, k, out=out)
    c, ms = ctx.timing_read(_native.K_BLUR_OP)
    b = 6 * h * w * n
    print("blur k=%-2d: %.1f us/launch  %.0f GB/s  %.0f frames/s" % (k, ms / c * 1e3, b / (ms / c * 1e-3) / 1e9, n / (ms / c * 1e-3)))
for (dw, dh) in ((426, 240), (960, 540), (1280, 720)):
    out = ctx.resize(frames, dw, dh)
    ctx.timing_enable([_native.K_RESIZE]); ctx.timing_reset()
    for _ in range(10):
        ctx.resize(frames, dw, dh, out=out)
    c, ms = ctx.timing_read(_native.K_RESIZE)
    print("resize 1080p -> %dx%d: %.1f us/launch  %.0f frames/s  (%.0f GB/s of source frames)" %
          (dw, dh, ms / c * 1e3, n / (ms / c * 1e-3), 3 * h * w * n / (ms / c * 1e-3) / 1e9))
# ConvertColor: bytes in + bytes out per launch against the HBM roofline
import numpy as np
from scannertools_amd._native import COLOR_CODES
nv12 = torch.randint(0, 256, (n, h * 3 // 2, w, 1), dtype=torch.uint8, device="cuda", generator=g)
uyvy = torch.randint(0, 256, (n, h, w, 2), dtype=torch.uint8, device="cuda", generator=g)
rgba = torch.randint(0, 256, (n, h, w, 4), dtype=torch.uint8, device="cuda", generator=g)
for name, src in (("COLOR_RGB2BGR", frames), ("COLOR_RGB2GRAY", frames), ("COLOR_RGB2HSV", frames), ("COLOR_HSV2RGB", frames),
                  ("COLOR_RGB2YCrCb", frames), ("COLOR_RGB2XYZ", frames), ("COLOR_RGB2RGBA", frames), ("COLOR_RGBA2BGR565", rgba),
                  ("COLOR_YUV2RGB_NV12", nv12), ("COLOR_YUV2BGR_I420", nv12), ("COLOR_YUV2RGB_UYVY", uyvy)):
    out = ctx.cvt_color(src, name)
    ctx.timing_enable([_native.K_CVT_COLOR]); ctx.timing_reset()
    for _ in range(10):
        ctx.cvt_color(src, name, out=out)
    c, ms = ctx.timing_read(_native.K_CVT_COLOR)
    b = src.numel() + out.numel()
    print("cvt %-20s %.1f us/launch  %.0f GB/s  %.0f frames/s" % (name, ms / c * 1e3, b / (ms / c * 1e-3) / 1e9, n / (ms / c * 1e-3)))
