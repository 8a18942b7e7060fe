"""Time individual Farneback stage kernels at 1080p x 33 frames through st_farneback_pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scannertools_amd import _native
from scannertools_amd.hip import HipContext
ctx = HipContext(0)
g = torch.Generator(device="cuda").manual_seed(0)
NF = int(os.environ.get("NF", 33))
fr = torch.randint(0, 256, (NF, 1080, 1920, 3), dtype=torch.uint8, device="cuda", generator=g)
out = ctx.optical_flow(fr)
ids = [_native.K_GRAY, _native.K_PYR, _native.K_POLYEXP, _native.K_BLUR_UPDATE]
ctx.timing_enable(ids); ctx.timing_reset()
for _ in range(5):
    ctx.optical_flow(fr, out=out)
for k in ids:
    n, ms = ctx.timing_read(k)
    print("%-16s %3d launches  %.3f ms per step" % (_native.KERNEL_NAMES[k], n, ms / 5))
