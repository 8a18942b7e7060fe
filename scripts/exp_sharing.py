"""Does k_flow_iter benefit from consecutive pairs sharing a frame's expansion in L2/MALL?
Times 256 consecutive pairs (257 frames) against 256 disjoint pairs (512 frames)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scannertools_amd import _native
from scannertools_amd.hip import HipContext
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_stream
ctx = HipContext(0)
dev = torch.device("cuda", 0)
P = int(os.environ.get("P", 256))
frames = make_stream(torch, dev, 2 * P, 1080, 1920, seed=1)
out = torch.empty((P, 1080, 1920, 2), dtype=torch.float32, device=dev)
for name, pairs in (("consecutive", [(i, i + 1) for i in range(P)]), ("disjoint", [(2 * i, 2 * i + 1) for i in range(P)]),
                    ("same frame twice", [(i, i) for i in range(P)])):
    ctx.optical_flow(frames, pairs=pairs, out=out)
    ctx.timing_enable([_native.K_BLUR_UPDATE, _native.K_POLYEXP]); ctx.timing_reset()
    for _ in range(3):
        ctx.optical_flow(frames, pairs=pairs, out=out)
    n, ms = ctx.timing_read(_native.K_BLUR_UPDATE)
    n2, ms2 = ctx.timing_read(_native.K_POLYEXP)
    print("%-18s flow_iter %.3f ms per step (%d launches)   polyexp %.3f ms per step" % (name, ms / 3, n // 3, ms2 / 3))
