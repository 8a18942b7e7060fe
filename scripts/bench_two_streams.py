"""(argv[1]: optional stagger in ms between the instances' starts.)  Does splitting a large OpticalFlow call over two kernel instances (two contexts, two streams, two host threads) fill
the tails of each other's launches?  256 pairs of 1080p as 1 x 256, 2 x 128 and 4 x 64 concurrent calls.
    python scripts/bench_two_streams.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from scannertools_amd.hip import HipContext

dev = torch.device("cuda", 0)
h, w, total = 1080, 1920, 256
fr = bench.make_stream(torch, dev, total + 1, h, w, seed=1)
out = torch.empty((total, h, w, 2), dtype=torch.float32, device=dev)
import sys as _sys
STAGGER_MS = float(_sys.argv[1]) if len(_sys.argv) > 1 else 0.0   # instance k starts k * STAGGER_MS late (complementary phases)
for K in (1, 2, 4, 1, 2, 4):
    b = total // K
    ctxs = [HipContext(0) for _ in range(K)]
    reps = 6
    barrier, done = threading.Barrier(K + 1), [0.0] * K

    def worker(k):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for i in range(reps + 2):
                if i == 2:
                    s.synchronize()
                    barrier.wait()
                    if STAGGER_MS:
                        time.sleep(k * STAGGER_MS * 1e-3)
                ctxs[k].optical_flow(fr[k * b:(k + 1) * b + 1], out=out[k * b:(k + 1) * b])
            s.synchronize()
        done[k] = time.perf_counter()

    th = [threading.Thread(target=worker, args=(k,)) for k in range(K)]
    for t in th:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = max(done) - t0
    print("%d instance(s) x %3d pairs per call: %6.0f frames/s, %.2f ms per 256 pairs" % (K, b, total * reps / dt, dt / reps * 1e3), flush=True)
    for c in ctxs:
        c.close()
