"""GPU: the threading contract of the boundary.

Scanner runs several instances of a kernel class concurrently in one process -- ``pipeline_instances_per_node`` > 1
(scannertools/tests/test_all.py:45,231) -- each from its own evaluator thread, each selecting its device at the top of
every method (optical_flow_kernel_gpu.cpp:18,29,37,41,47; histogram_kernel_gpu.cpp:21,35).  include/scannertools_hip.h
promises the same of this library: one ``st_ctx`` per kernel instance, entry points re-entrant across contexts.  Here K host
threads drive K contexts (and K instances of the op library's kernel classes) on ONE GPU at the same time, on different
inputs, and every output must be bit-identical to what a single context produces alone.
"""
import threading

import numpy as np
import pytest
import torch

from scannertools_amd.hip import HipContext
from util import texture_stream

pytestmark = pytest.mark.gpu

K = 4


def _streams(h, w, n, seed0):
    return [torch.from_numpy(texture_stream(seed0 + k, n, h, w)[0]).cuda() for k in range(K)]


def _work(ctx, frames, reps):
    """The calls one kernel instance makes: Histogram over its frames and OpticalFlow with 1, 2 and all pairs per call."""
    out = []
    n = len(frames)
    for r in range(reps):
        ppc = (1, 2, n - 1)[r % 3]
        j = r % (n - ppc)
        out.append(ctx.histogram(frames, 16 if r % 2 == 0 else 256).clone())
        out.append(ctx.optical_flow(frames[j:j + ppc + 1]).clone())
    return out


def _run_threads(fn):
    """fn(k) on K threads released together; re-raises the first exception."""
    barrier, errors, results = threading.Barrier(K), [], [None] * K

    def body(k):
        try:
            barrier.wait()
            results[k] = fn(k)
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append(e)
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=body, args=(k,)) for k in range(K)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a worker thread hangs"
    if errors:
        raise errors[0]
    return results


@pytest.mark.parametrize("h,w,n,reps", [(270, 480, 5, 9), (1080, 1920, 3, 4)])
def test_four_contexts_on_four_threads_match_the_serial_run(h, w, n, reps):
    """K threads, each with its own st_ctx on its own stream, all running Histogram + OpticalFlow at once on different
    streams of frames: every result equals the serial run's, bit for bit (scratch, tables and streams are per context;
    nothing global is written after library load)."""
    streams = _streams(h, w, n, 300)
    with HipContext(0) as ref_ctx:
        serial = [[t.cpu() for t in _work(ref_ctx, streams[k], reps)] for k in range(K)]
        torch.cuda.synchronize()
    ctxs = [HipContext(0) for _ in range(K)]

    def fn(k):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            res = _work(ctxs[k], streams[k], reps)
            s.synchronize()
        return [t.cpu() for t in res]

    try:
        for _ in range(2):      # second round: scratch already sized, the calls overlap from their first launch
            got = _run_threads(fn)
            for k in range(K):
                assert len(got[k]) == len(serial[k])
                for a, b in zip(got[k], serial[k]):
                    assert a.dtype == b.dtype and torch.equal(a, b), "instance %d differs from the serial run" % k
    finally:
        for c in ctxs:
            c.close()


def test_contexts_created_and_destroyed_while_others_run():
    """Kernel instances come and go while others execute (Scanner tears instances down per task group): context creation,
    workspace growth and destruction on one thread must not disturb the launches of another."""
    streams = _streams(216, 384, 4, 340)
    with HipContext(0) as ref_ctx:
        serial = [[t.cpu() for t in _work(ref_ctx, streams[k], 3)] for k in range(K)]

    def fn(k):
        res = []
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                with HipContext(0) as ctx:
                    res = _work(ctx, streams[k], 3)
                    s.synchronize()
        return [t.cpu() for t in res]

    got = _run_threads(fn)
    for k in range(K):
        for a, b in zip(got[k], serial[k]):
            assert torch.equal(a, b)


def test_two_contexts_on_one_torch_stream():
    """Two contexts bound (st_ctx_set_stream) to the SAME torch stream: their launches interleave in stream order, from one
    thread and from two; each keeps its own scratch, so the results are the serial ones."""
    fa, fb = _streams(270, 480, 5, 360)[:2]
    with HipContext(0) as ref_ctx:
        want_a = [t.cpu() for t in _work(ref_ctx, fa, 3)]
        want_b = [t.cpu() for t in _work(ref_ctx, fb, 3)]
    shared = torch.cuda.Stream()
    with HipContext(0) as ca, HipContext(0) as cb:
        with torch.cuda.stream(shared):
            got_a, got_b = [], []
            for r in range(3):                      # interleaved call by call, no synchronisation in between
                ppc = (1, 2, 4)[r]
                got_a.append(ca.histogram(fa, 16 if r % 2 == 0 else 256).clone())
                got_b.append(cb.histogram(fb, 16 if r % 2 == 0 else 256).clone())
                got_a.append(ca.optical_flow(fa[r % (5 - ppc):r % (5 - ppc) + ppc + 1]).clone())
                got_b.append(cb.optical_flow(fb[r % (5 - ppc):r % (5 - ppc) + ppc + 1]).clone())
            shared.synchronize()
        for got, want in ((got_a, want_a), (got_b, want_b)):
            for a, b in zip(got, want):
                assert torch.equal(a.cpu(), b)
        # ... and from two threads enqueueing on that one stream at the same time
        res = {}

        def worker(name, ctx, frames):
            with torch.cuda.stream(shared):
                res[name] = _work(ctx, frames, 3)

        ta = threading.Thread(target=worker, args=("a", ca, fa))
        tb = threading.Thread(target=worker, args=("b", cb, fb))
        ta.start(); tb.start(); ta.join(600); tb.join(600)
        shared.synchronize()
        for got, want in ((res["a"], want_a), (res["b"], want_b)):
            for a, b in zip(got, want):
                assert torch.equal(a.cpu(), b)


def test_four_instances_of_the_op_library_kernel_classes():
    """The same through the Scanner-style kernel classes: four threads, each building and running its own graph
    (Histogram + OpticalFlow on DeviceType.GPU -> one HistogramKernelHIP and one OpticalFlowKernelHIP instance per thread,
    one pair / one frame per execute() and batched), all on one GPU at once; rows equal the serial run's."""
    from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, NamedVideoStream, PerfParams
    streams = _streams(270, 480, 6, 380)

    def graph(k, batch):
        sc = Client()
        sc.ingest_frames("v", streams[k])
        frame = sc.io.Input([NamedVideoStream(sc, "v")])
        kw = {} if batch is None else {"batch": batch}
        hist, flow = NamedStream(sc, "hist"), NamedStream(sc, "flow")
        sc.run([sc.io.Output(sc.ops.Histogram(frame=frame, device=DeviceType.GPU, **kw), [hist]),
                sc.io.Output(sc.ops.OpticalFlow(frame=frame, device=DeviceType.GPU, **kw), [flow])],
               PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
        return [np.stack(h) for h in hist.load()], [np.asarray(f) for f in flow.load()]

    for batch in (None, 4):
        serial = [graph(k, batch) for k in range(K)]
        got = _run_threads(lambda k: graph(k, batch))
        for k in range(K):
            for a, b in zip(got[k][0], serial[k][0]):
                np.testing.assert_array_equal(a, b)
            assert len(got[k][1]) == 6
            for a, b in zip(got[k][1], serial[k][1]):
                np.testing.assert_array_equal(a, b)


def test_small_calls_notice_other_instances_and_results_do_not_change():
    """Kernel choice under concurrency (st_ctx_flow_concurrent): a small OpticalFlow call that finds TWO other contexts of the
    process with a call in flight picks kernels that share a CU (no role-split workgroups); alone, beside one other instance,
    with more than 4 pairs, or once the others have synchronised it does not.  The flows are the same bits either way."""
    h, w = 256, 320
    frames = torch.from_numpy(texture_stream(41, 10, h, w)[0]).cuda()
    a, b, c = HipContext(0), HipContext(0), HipContext(0)
    try:
        alone = c.optical_flow(frames[:3]).clone()
        assert not c.flow_concurrent()
        c.sync()
        a.optical_flow(frames[2:5])                 # in flight, not synchronised
        beside_one = c.optical_flow(frames[:3]).clone()
        assert not c.flow_concurrent()
        c.sync()
        b.optical_flow(frames[4:7])                 # a second one in flight
        shared = c.optical_flow(frames[:3]).clone()
        assert c.flow_concurrent()
        big = c.optical_flow(frames[:9])            # 8 pairs: fills the chip by itself
        assert not c.flow_concurrent()
        a.sync(); b.sync(); c.sync()
        after = c.optical_flow(frames[:3]).clone()
        assert not c.flow_concurrent()
        c.sync()
        assert torch.equal(alone, beside_one) and torch.equal(alone, shared) and torch.equal(alone, after)
        assert torch.equal(big[:2], alone)
    finally:
        for x in (a, b, c):
            x.close()
