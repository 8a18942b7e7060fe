"""Bit-exactness campaign of the flow path's scheduling modes (not part of the test-suite): random frame sizes and pair
lists, the flow of every mode -- kernel by launch size, marching, tile, role-split with 4 and 5 column waves -- against the
marching kernel's, bit for bit.   python scripts/fuzz_schedules.py [n_seeds] [max_h] [max_w]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from conftest import FLOW_MODES, make_mode_ctx
from util import smooth_texture

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
max_h = int(sys.argv[2]) if len(sys.argv) > 2 else 400
max_w = int(sys.argv[3]) if len(sys.argv) > 3 else 700
ctxs = {m: make_mode_ctx(m) for m in FLOW_MODES}
bad = fields = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(9000 + seed)
    h, w = int(rng.integers(2, max_h)), int(rng.integers(2, max_w))
    if seed % 10 == 0:
        h, w = int(rng.integers(400, 1100)), int(rng.integers(600, 2000))   # sizes whose levels march
    nf = int(rng.integers(2, 6))
    base = np.stack([smooth_texture(int(rng.integers(1 << 30)), h + 8, w + 8, sigma=float(rng.choice([1.5, 3.0, 8.0]))) for _ in range(3)], -1)
    frames = np.stack([base[dy:dy + h, dx:dx + w] for dy, dx in rng.integers(0, 9, (nf, 2))]).astype(np.uint8)
    pairs = [(int(a), int(b)) for a, b in rng.integers(0, nf, (int(rng.integers(1, 7)), 2))]
    d = torch.from_numpy(frames).cuda()
    ref = ctxs["march"].optical_flow(d, pairs=pairs)
    fields += len(pairs)
    for m, c in ctxs.items():
        if m == "march":
            continue
        got = c.optical_flow(d, pairs=pairs)
        if not torch.equal(got, ref):
            bad += 1
            print("MISMATCH seed %d mode %s size %dx%d pairs %s max|diff| %g" % (seed, m, h, w, pairs, float((got - ref).abs().max())), flush=True)
print("seeds %d, flow fields %d, modes %s: mismatches %d" % (n_seeds, fields, sorted(ctxs), bad))
sys.exit(1 if bad else 0)
