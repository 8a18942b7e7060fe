"""Per-kernel-class time of one headline step (257 frames of 1080p -> 256 flow fields + histograms) by the library's HIP-event
brackets: python scripts/time_kernels.py [steps]   (ST_HIP_LIB selects a variant build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from scannertools_amd import _native
from scannertools_amd.hip import HipContext
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
fr = bench.make_stream(torch, dev, 257, 1080, 1920, seed=1)
with HipContext(0) as ctx:
    flow = torch.empty((256, 1080, 1920, 2), dtype=torch.float32, device=dev)
    for _ in range(2):
        ctx.optical_flow(fr, out=flow)
    ids = [_native.K_GRAY, _native.K_PYR, _native.K_POLYEXP, _native.K_BLUR_UPDATE]
    ctx.timing_enable(ids); ctx.timing_reset()
    for _ in range(steps):
        ctx.optical_flow(fr, out=flow)
    torch.cuda.synchronize()
    print(os.environ.get("ST_HIP_LIB", "default"), {_native.KERNEL_NAMES[i]: round(ctx.timing_read(i)[1] / steps, 3) for i in ids})
