"""bench.py's `extra.concurrent_instances` record on its own (K kernel instances x b pairs per call, a st_ctx_sync after every
call as the kernel classes' execute() does): python scripts/bench_concurrent.py   (ST_CONCURRENT=0 switches the detection of
other instances off: every instance then picks its kernels as if it were alone)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
frames = bench.make_stream(torch, dev, 129, 1080, 1920, seed=1)
rec = bench.concurrent_instances(torch, dev, frames, 1080, 1920, 256, pairs=(1, 2, 8))
for b in (1, 2, 8):
    row = rec["pairs_per_call_%d" % b]
    print("pairs per call %d: " % b + "  ".join("K=%s %.0f [%.0f, %.0f]" % (k[2:], v["frames_per_s"], *v["frames_per_s_range"]) for k, v in row.items()))
