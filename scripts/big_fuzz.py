"""Long differential fuzz run of tests/test_fuzz_gpu.py's generators (integer ops, flows, pose ops); not part of the
test-suite.  python scripts/big_fuzz.py [n_int] [n_flow] [n_pose]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ["ST_RECORD_FLOW_TIERS"] = "1"   # campaign mode: every seed reports its tiers instead of being checked against the pinned list
import test_fuzz_gpu as F
from scannertools_amd.hip import HipContext
ctx = HipContext(0)
bad = 0
n_int, n_flow, n_pose = (int(v) for v in (sys.argv[1:4] + ['160', '40', '60'][len(sys.argv) - 1:]))
for seed in range(6, n_int):
    try:
        F.test_fuzz_integer_ops(ctx, seed)
    except AssertionError as e:
        bad += 1; print("INT FAIL seed", seed, str(e)[:300])
for seed in range(3, n_flow):
    try:
        F.test_fuzz_optical_flow(ctx, seed)
    except AssertionError as e:
        bad += 1; print("FLOW FAIL seed", seed, str(e)[:300])
for seed in range(4, n_pose):
    try:
        F.test_fuzz_pose_ops(ctx, seed)
    except AssertionError as e:
        bad += 1; print("POSE FAIL seed", seed, str(e)[:300])
print("flow fields by tolerance tier (1 plain, 2 noise-arbitrated, 3 branch flip):", F.FLOW_TIERS)
print("done, failures:", bad)
