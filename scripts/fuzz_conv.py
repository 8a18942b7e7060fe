"""Bit-exactness campaign of the pose convolution kernels (not part of the test-suite): random map sizes, channel counts, batch
sizes and channel slices; every kernel choice (ST_CONV_TILE = 0 per-tap, 1 eight-wave tile, 4 four-wave tile, 41 one instruction
tile per wave, unset: by launch size) must give the same bits in each arithmetic, and float64 agreement within the tests' bound.
  python scripts/fuzz_conv.py [n_cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import test_pose_net_gpu as T
from scannertools_amd.hip import HipContext

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
auto = HipContext(0)
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(77000 + case)
    k = (3, 7)[case % 2]
    h, w = int(rng.integers(1, 120)), int(rng.integers(1, 200))
    ci = int(rng.choice([16, 32, 64, 96]))
    co = int(rng.choice([128, 97, 256, 130, 64, 19]))
    n = int(rng.integers(1, 7))
    xoff = int(rng.choice([0, 16]))
    g = torch.Generator().manual_seed(case)
    x = torch.randn((n, h, w, ci + xoff + 16), generator=g)
    wt = torch.randn((co, ci, k, k), generator=g) * float(np.sqrt(2.0 / (ci * k * k)))
    b = torch.randn((co,), generator=g) * 0.1
    ref = torch.relu(torch.nn.functional.conv2d(x[..., xoff:xoff + ci].permute(0, 3, 1, 2).double(), wt.double(), b.double(), padding=k // 2))
    ys = (co + 3) // 4 * 4 + 8
    for math in ("f32", "bf16x3"):
        outs = {}
        for mode in ("", "_pertap", "_tile4", "_tile41"):
            outs[mode] = T._conv(auto, x.cuda(), ci, xoff, wt, b, 1, cout_total=ys, yoff=4, math=math + mode)
        # `_conv` maps the plain names to the forced 8-wave context; the by-launch-size choice runs through the pair entry in the tests
        base = outs[""]
        got = base[..., 4:4 + co].permute(0, 3, 1, 2).cpu().double()
        scale = max(float(ref.abs().max()), 1.0)
        ok = float((got - ref).abs().max()) <= 2e-5 * scale and all(torch.equal(base, v) for v in outs.values())
        ok = ok and bool((base[..., :4] == -7.0).all()) and bool((base[..., 4 + co:] == -7.0).all())
        if not ok:
            bad += 1
            print("FAIL case", case, math, (n, h, w, ci, co, k, xoff), {m: float((base - v).abs().max()) for m, v in outs.items()})
print("cases", n_cases, "x 2 arithmetics x 4 kernel choices; failures:", bad)
