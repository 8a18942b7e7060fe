"""Campaign: OpticalFlow (HIP, C ABI) against the oracle on motion that is not a whole-pixel shift -- every kind of
tests/util.py: motion_pair at random frame sizes and seeds, under a scheduling mode drawn per case; each field through
util.assert_flow_close (tier 1 plain bound, 2 noise-arbitrated, 3 branch flip).  Not part of the test-suite.
    python scripts/fuzz_motion.py [n_cases] [seed0]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import oracle
from conftest import FLOW_MODES, make_mode_ctx
from util import MOTION_KINDS, assert_flow_close, motion_pair

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 140
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctxs = {m: make_mode_ctx(m) for m in FLOW_MODES}
modes = sorted(FLOW_MODES)
tiers, worst, bad = collections.Counter(), {}, 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    kind = MOTION_KINDS[case % len(MOTION_KINDS)]
    h, w = int(rng.integers(100, 560)), int(rng.integers(130, 760))
    if case % 5 == 0:
        h, w = 8 * (h // 8 + 30), 8 * (w // 8 + 30)      # a one-pass-pyramid geometry
    mode = modes[int(rng.integers(len(modes)))]
    a, b = motion_pair(kind, seed0 + case, h, w)
    ref = oracle.optical_flow_rgb(a, b)
    got = ctxs[mode].optical_flow(torch.from_numpy(np.stack([a, b])).cuda()).cpu().numpy()[0]
    try:
        t = assert_flow_close(got, ref, a, b, (kind, h, w, mode))
        tiers[t] += 1
        rel = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))
        mx = float(np.abs(got - ref).max())
        if kind not in worst or mx > worst[kind][1]:
            worst[kind] = (rel, mx, h, w, mode, t)
    except AssertionError as e:
        bad += 1
        print("FAIL case %d %s %dx%d %s: %s" % (case, kind, h, w, mode, str(e)[:300]), flush=True)
print("motion campaign: %d cases (seeds %d..%d), tiers %s, failures %d" % (n_cases, seed0, seed0 + n_cases - 1, dict(tiers), bad))
for k in MOTION_KINDS:
    if k in worst:
        print("  worst %-10s rel-L2 %.3g max-abs %.3g px at %dx%d under '%s' (tier %d)" % ((k,) + worst[k]))
sys.exit(1 if bad else 0)
