"""Differential campaign for the gray-source level-0 expansion (k_polyexp_u8 + k_pyr_roles<false>, the default for calls above 16
pairs on one-pass-pyramid geometries) against the float-source schedule (ST_POLY_U8=0): random frame sizes (multiples of 8 from
256 up: one to several strips, 8-column tail strips, several segments), random pair counts above 16, three kinds of content
(uniform random bytes, a smooth texture under translation, flat frames with a few impulses -- every reflected border weight and
clamped row shows in one of them).  Bit-identical flows required.   python scripts/fuzz_polyu8.py [n_cases] [seed0]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from conftest import make_mode_ctx

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
a, b = make_mode_ctx("polyu8"), make_mode_ctx("polyf32")
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    h = 8 * int(rng.integers(32, 90))
    w = 8 * int(rng.integers(32, 200))
    n = int(rng.integers(18, 27))
    kind = case % 3
    g = torch.Generator(device="cuda").manual_seed(seed0 + case)
    if kind == 0:
        fr = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    elif kind == 1:
        low = torch.rand((1, 3, h // 8 + 12, w // 8 + 12), device="cuda", generator=g)
        tex = torch.nn.functional.interpolate(low, size=(h + 64, w + 64), mode="bicubic", align_corners=False)[0]
        tex = ((tex - tex.amin()) / (tex.amax() - tex.amin()) * 255).permute(1, 2, 0)
        fr = torch.stack([tex[32 + (i % 5):32 + (i % 5) + h, 32 - (i % 7):32 - (i % 7) + w] for i in range(n)]).to(torch.uint8).contiguous()
    else:
        fr = torch.full((n, h, w, 3), int(rng.integers(0, 256)), dtype=torch.uint8, device="cuda")
        for i in range(n):
            for _ in range(40):
                y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
                fr[i, y if rng.random() < 0.7 else int(rng.choice([0, 1, h - 2, h - 1])), x if rng.random() < 0.7 else int(rng.choice([0, 1, w - 2, w - 1]))] = int(rng.integers(0, 256))
    fa, fb = a.optical_flow(fr), b.optical_flow(fr)
    ok = torch.equal(fa, fb)
    if not ok:
        bad += 1
        print("MISMATCH case %d: %dx%d n=%d kind=%d max|d|=%g" % (case, h, w, n, kind, float((fa - fb).abs().max())), flush=True)
    del fr, fa, fb
print("polyu8 campaign: %d cases (seeds %d..%d), mismatches: %d" % (n_cases, seed0, seed0 + n_cases - 1, bad))
sys.exit(1 if bad else 0)
