"""Long differential fuzz run (154 + 37 seeds of tests/test_fuzz_gpu.py); not part of the test-suite."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import test_fuzz_gpu as F
from scannertools_amd.hip import HipContext
ctx = HipContext(0)
bad = 0
for seed in range(6, 160):
    try:
        F.test_fuzz_integer_ops(ctx, seed)
    except AssertionError as e:
        bad += 1; print("INT FAIL seed", seed, str(e)[:300])
for seed in range(3, 40):
    try:
        F.test_fuzz_optical_flow(ctx, seed)
    except AssertionError as e:
        bad += 1; print("FLOW FAIL seed", seed, str(e)[:300])
print("done, failures:", bad)
