"""Histogram-only micro benchmark: random, smooth and all-equal 1080p frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scannertools_amd import _native
from scannertools_amd.hip import HipContext
ctx = HipContext(0)
n, h, w = int(os.environ.get('N', 64)), 1080, 1920
g = torch.Generator(device="cuda").manual_seed(0)
data = {"random": torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=g),
        "equal": torch.full((n, h, w, 3), 77, dtype=torch.uint8, device="cuda")}
sm = torch.nn.functional.interpolate(torch.rand((n, 3, h // 16, w // 16), device="cuda", generator=g), size=(h, w), mode="bilinear")
data["smooth"] = (sm.permute(0, 2, 3, 1) * 255).to(torch.uint8).contiguous()
for name, fr in data.items():
    for bins in (256, 16):
        out = ctx.histogram(fr, bins)
        ref = torch.stack([torch.bincount(fr[0, ..., c].flatten().int() * bins // 256, minlength=bins) for c in range(3)])
        assert torch.equal(out[0].long(), ref), name
        ctx.timing_enable([_native.K_HIST]); ctx.timing_reset()
        for _ in range(20):
            ctx.histogram(fr, bins, out=out)
        k, ms = ctx.timing_read(_native.K_HIST)
        bytes_ = (3 * h * w + 3 * bins * 4) * n
        print("%-7s bins %3d: %.1f us/launch  %.0f GB/s  %.0f frames/s" % (name, bins, ms / k * 1e3, bytes_ / (ms / k * 1e-3) / 1e9, n / (ms / k * 1e-3)))
