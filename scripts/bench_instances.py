"""Unbatched OpticalFlow calls (one pair per call, what an unchanged graph with the reference's un-batched CPU
registration drives) from K concurrent kernel instances -- K contexts, K streams, K host threads, which is how
Scanner runs K pipeline instances of an op on one GPU.  Prints pairs/s against K.
    python scripts/bench_instances.py [pairs_per_call]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scannertools_amd.hip import HipContext
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import texture_stream

ppc = int(sys.argv[1]) if len(sys.argv) > 1 else 1
h, w = 1080, 1920
frames = torch.from_numpy(texture_stream(3, 33, h, w)[0]).cuda()
calls = 48


def worker(k, ctx, out, barrier, done):
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx.optical_flow(frames[:ppc + 1], out=out)  # warm-up: scratch allocation
        stream.synchronize()
        barrier.wait()
        for i in range(calls):
            j = (i * ppc + k) % (32 - ppc)
            ctx.optical_flow(frames[j:j + ppc + 1], out=out)
        stream.synchronize()
    done[k] = time.perf_counter()


for K in (1, 2, 4, 8, 16, 32):
    ctxs = [HipContext(0) for _ in range(K)]
    outs = [torch.empty((ppc, h, w, 2), dtype=torch.float32, device="cuda") for _ in range(K)]
    barrier, done = threading.Barrier(K + 1), [0.0] * K
    th = [threading.Thread(target=worker, args=(k, ctxs[k], outs[k], barrier, done)) for k in range(K)]
    for t in th:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = max(done) - t0
    print("%2d instances x %d pair(s) per call: %7.0f frames/s" % (K, ppc, K * calls * ppc / dt))
    del ctxs, outs
    torch.cuda.empty_cache()
