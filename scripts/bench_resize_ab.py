import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scannertools_amd import _native
from scannertools_amd.hip import HipContext
ctx = HipContext(0)
n, h, w = 64, 1080, 1920
frames = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
for (dw, dh) in ((1280, 720), (1278, 720), (640, 360), (1600, 900), (428, 240), (426, 240)):
    for interp in (1,):
        out = ctx.resize(frames, dw, dh, interp)
        ctx.timing_enable([_native.K_RESIZE]); ctx.timing_reset()
        for _ in range(10):
            ctx.resize(frames, dw, dh, interp, out=out)
        c, ms = ctx.timing_read(_native.K_RESIZE)
        print("resize 1080p -> %dx%d: %.1f us/launch  %.0f frames/s  out %.0f GB/s" % (dw, dh, ms / c * 1e3, n / (ms / c * 1e-3), 3 * dw * dh * n / (ms / c * 1e-3) / 1e9))
