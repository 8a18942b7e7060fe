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
for k in (3, 5, 15):
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
