"""Resize op, every interpolation mode at two targets (64 x 1080p, 3 channels): us per launch and frames/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scannertools_amd import _native
from scannertools_amd.hip import HipContext
ctx = HipContext(0)
n, h, w = 64, 1080, 1920
frames = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
names = {0: "NEAREST", 1: "LINEAR", 2: "CUBIC", 3: "AREA", 4: "LANCZOS4"}
for (dw, dh) in ((1280, 720), (426, 240), (640, 360), (2560, 1440)):
    for interp in (0, 1, 2, 3, 4):
        out = ctx.resize(frames, dw, dh, interp)
        ctx.timing_enable([_native.K_RESIZE]); ctx.timing_reset()
        for _ in range(5):
            ctx.resize(frames, dw, dh, interp, out=out)
        c, ms = ctx.timing_read(_native.K_RESIZE)
        print("1080p -> %4dx%-4d %-8s %8.1f us/launch  %9.0f frames/s" % (dw, dh, names[interp], ms / c * 1e3, n / (ms / c * 1e-3)))
