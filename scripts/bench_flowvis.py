"""FlowHistogram / DrawFlow micro benchmark on device-resident 1080p flow frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scannertools_amd import _native
from scannertools_amd.hip import HipContext

ctx = HipContext(0)
n, h, w = int(os.environ.get("N", 64)), 1080, 1920
g = torch.Generator(device="cuda").manual_seed(0)
flows = {"random": torch.randn((n, h, w, 2), device="cuda", generator=g) * 6,
         "small": torch.randn((n, h, w, 2), device="cuda", generator=g) * 0.2,   # every magnitude in bin 0
         "zero": torch.zeros((n, h, w, 2), device="cuda")}
frames = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
for name, fl in flows.items():
    out = ctx.flow_histogram(fl)
    ctx.timing_enable([_native.K_FLOW_HIST]); ctx.timing_reset()
    for _ in range(20):
        ctx.flow_histogram(fl, out=out)
    k, ms = ctx.timing_read(_native.K_FLOW_HIST)
    b = (8 * h * w + 512) * n
    print("flow_hist %-6s: %.1f us/launch  %.0f GB/s  %.0f frames/s" % (name, ms / k * 1e3, b / (ms / k * 1e-3) / 1e9, n / (ms / k * 1e-3)))
fl = flows["random"]
out = ctx.draw_flow(frames, fl)
ctx.timing_enable([_native.K_DRAW_FLOW]); ctx.timing_reset()
for _ in range(20):
    ctx.draw_flow(frames, fl, out=out)
k, ms = ctx.timing_read(_native.K_DRAW_FLOW)
b = (8 + 8 + 3 + 6) * h * w * n   # flow twice (max pass + render), frame in, picture out
print("draw_flow       : %.1f us/call  %.0f GB/s  %.0f frames/s" % (ms / k * 1e3, b / (ms / k * 1e-3) / 1e9, n / (ms / k * 1e-3)))
