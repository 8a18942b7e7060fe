"""ShotBoundaries on the device (st_shot_boundaries) against the golden lists produced by importing the reference's
shot_detection.py (tests/golden/make_shot_golden.py) and against the host op on random streams: the decisions are
float64 comparisons, so the device code reproduces numpy's summation order and must agree bit for bit."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "shot_golden.npz"))
CASES = sorted({k.rsplit("__", 1)[0] for k in GOLD.files})


def np_pairwise(a):
    """Pure-Python twin of the device routine `np_pairwise` (csrc/st_hist.hip): numpy's pairwise summation of a
    contiguous float64 vector (numpy/_core/src/umath/loops_utils.h.src)."""
    n = len(a)
    if n < 8:
        r = 0.0
        for v in a:
            r += v
        return r
    if n <= 128:
        r = [a[k] for k in range(8)]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] += a[i + k]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res += a[i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return np_pairwise(a[:n2]) + np_pairwise(a[n2:])


def test_pairwise_emulation_is_numpys_summation():
    """The summation order the device code follows IS numpy's: sum, mean and std of float64 vectors of every length
    that occurs as a window (1 .. 1000) and beyond, bit for bit, on the numpy this suite runs with."""
    rng = np.random.default_rng(0)
    for n in list(range(1, 200)) + [248, 255, 256, 257, 499, 500, 501, 640, 999, 1000, 1001, 1537, 4096]:
        a = rng.random(n) * float(rng.choice([1.0, 1e3, 1e-3]))
        al = [float(v) for v in a]
        s = np_pairwise(al)
        assert s == float(np.add.reduce(a)) == float(np.sum(a)), n
        mean = s / n
        assert mean == float(np.mean(a)), n
        sq = [(v - mean) * (v - mean) for v in al]
        assert np.sqrt(np_pairwise(sq) / n) == float(np.std(a)), n


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_device_shot_boundaries_match_the_reference_goldens(hip_ctx, case):
    import torch
    h = np.ascontiguousarray(GOLD[case + "__hist"]).astype(np.int32)
    got = hip_ctx.shot_boundaries(torch.from_numpy(h).cuda())
    assert got == GOLD[case + "__bounds"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,bins", [(0, 1, 16), (1, 2, 16), (2, 3, 7), (3, 499, 16), (4, 1000, 16), (5, 1001, 256), (6, 2503, 16),
                                         (7, 10000, 16), (8, 30011, 16)])
def test_device_shot_boundaries_equal_the_host_op(hip_ctx, seed, n, bins):
    """Random histogram streams with planted cuts and noise of several magnitudes: distances (float64) and boundary
    lists of the device path equal the host op's exactly; windows other than the reference's too."""
    import torch
    from scannertools_amd import shot_detection
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 5000, (3, bins))
    h = np.empty((n, 3, bins), np.int32)
    for i in range(n):
        if rng.random() < 0.004:
            base = rng.integers(0, 5000, (3, bins))
        h[i] = base + rng.integers(-40, 41, (3, bins)) * int(rng.choice([1, 1, 5]))
    hd = torch.from_numpy(h).cuda()
    idx, diffs = hip_ctx.shot_boundaries(hd, return_diffs=True)
    ref_d = shot_detection.histogram_diffs(h)
    np.testing.assert_array_equal(diffs.cpu().numpy(), ref_d)
    assert idx == shot_detection.outlier_boundaries(ref_d)
    assert shot_detection.shot_boundaries_device(hip_ctx, hd) == shot_detection.shot_boundaries(None, list(h))
    # another window / threshold against a direct evaluation of the definition
    W, k = 37, 1.5
    got = hip_ctx.shot_boundaries(hd, window=W, k_std=k)
    want = [i for i in range(1, n) if ref_d[i] - np.mean(ref_d[max(i - W, 0):min(i + W, n)]) > k * np.std(ref_d[max(i - W, 0):min(i + W, n)])]
    assert got == want


@pytest.mark.gpu
def test_device_shot_boundaries_extremes(hip_ctx):
    """Counts at the ends of int32 (the distance needs 33 bits), a 200 000-frame stream, windows longer than the stream."""
    import torch
    from scannertools_amd import shot_detection
    rng = np.random.default_rng(11)
    h = rng.integers(0, 2, (300, 3, 16)).astype(np.int64) * (2 ** 32 - 1) - 2 ** 31     # INT32_MIN / INT32_MAX only
    h = h.astype(np.int32)
    hd = torch.from_numpy(h).cuda()
    idx, diffs = hip_ctx.shot_boundaries(hd, return_diffs=True)
    np.testing.assert_array_equal(diffs.cpu().numpy(), shot_detection.histogram_diffs(h))
    assert idx == shot_detection.outlier_boundaries(shot_detection.histogram_diffs(h))
    n = 200000
    level = np.cumsum(rng.random(n) < 2e-4)                      # ~40 scene changes
    base = ((level * 7919) % 13)[:, None, None] * 500
    h = (base + rng.integers(0, 60, (n, 3, 16))).astype(np.int32)
    got = hip_ctx.shot_boundaries(torch.from_numpy(h).cuda())
    assert got == shot_detection.shot_boundaries(None, list(h))[0] and len(got) > 10
    short = torch.from_numpy(h[:40]).cuda()
    whole = [i for i in range(1, 40) if (lambda d: d[i] - np.mean(d) > 2.5 * np.std(d))(shot_detection.histogram_diffs(h[:40]))]
    assert hip_ctx.shot_boundaries(short, window=5000) == whole
    assert hip_ctx.shot_boundaries(short, window=2 ** 31 - 1) == whole      # i + window must not overflow an int
    # a window of a whole 30 000-frame stream: every frame sums all of it (numpy's recursion 8 levels deep, evaluated with the
    # kernel's explicit stack; the work is quadratic in the stream, hence not the 200 000 frames), against numpy itself
    m = 30000
    d = shot_detection.histogram_diffs(h[:m])
    got = set(hip_ctx.shot_boundaries(torch.from_numpy(h[:m]).cuda(), window=m))
    want = set(int(i) for i in np.nonzero(d - np.mean(d) > 2.5 * np.std(d))[0] if i > 0)
    assert got == want and len(got) > 2


@pytest.mark.gpu
def test_shot_boundaries_of_an_empty_stream(hip_ctx):
    """No frames: the device function answers like the host op ([[]]), and the op on the GPU through the engine yields no rows."""
    import torch
    from scannertools_amd import shot_detection
    from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, NamedVideoStream, PerfParams
    assert shot_detection.shot_boundaries(None, []) == [[]]
    assert shot_detection.shot_boundaries_device(hip_ctx, torch.zeros((0, 3, 16), dtype=torch.int32, device="cuda")) == [[]]
    sc = Client()
    sc.ingest_frames("empty", np.zeros((0, 8, 8, 3), np.uint8))
    hist = sc.ops.Histogram(frame=sc.io.Input([NamedVideoStream(sc, "empty")]), device=DeviceType.CPU)
    for dev in (DeviceType.CPU, DeviceType.GPU):
        out = NamedStream(sc, "empty_sb_%d" % dev)
        sc.run(sc.io.Output(sc.ops.ShotBoundaries(histograms=hist, device=dev), [out]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
        assert list(out.load()) == []


@pytest.mark.gpu
def test_shot_boundaries_op_on_the_gpu_through_the_engine(hip_ctx):
    """sc.ops.ShotBoundaries(histograms=..., device=GPU) after the Histogram op: the same rows as the host op."""
    from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, NamedVideoStream, PerfParams
    from util import texture_stream
    frames, _ = texture_stream(3, 40, 48, 64)
    frames[17:] = 255 - frames[17:]       # a hard cut
    sc = Client()
    sc.ingest_frames("v", frames)
    frame = sc.io.Input([NamedVideoStream(sc, "v")])
    outs = {}
    for dev in (DeviceType.CPU, DeviceType.GPU):
        hist = sc.ops.Histogram(frame=frame, device=DeviceType.GPU, batch=8)
        o = NamedStream(sc, "sb%d" % int(dev == DeviceType.GPU))
        sc.run(sc.io.Output(sc.ops.ShotBoundaries(histograms=hist, device=dev), [o]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
        outs[dev] = list(o.load())
    assert outs[DeviceType.CPU] == outs[DeviceType.GPU] and 17 in outs[DeviceType.GPU][0] and outs[DeviceType.GPU][1] is None


@pytest.mark.gpu
def test_shot_boundaries_entry_point_refuses_bad_arguments(hip_ctx):
    import ctypes
    import torch
    from scannertools_amd import _native
    L, h = hip_ctx._L, hip_ctx._h
    hist = torch.zeros((4, 3, 16), dtype=torch.int32, device="cuda")
    fl = torch.zeros((4,), dtype=torch.uint8, device="cuda")
    vp = ctypes.c_void_p
    assert L.st_shot_boundaries(h, vp(hist.data_ptr()), 4, 16, 500, 2.5, vp(fl.data_ptr()), None) == 0
    assert L.st_shot_boundaries(h, vp(hist.data_ptr()), 0, 16, 500, 2.5, None, None) == 0
    for bad in ((None, 4, 16, 500, vp(fl.data_ptr())), (vp(hist.data_ptr()), -1, 16, 500, vp(fl.data_ptr())), (vp(hist.data_ptr()), 4, 0, 500, vp(fl.data_ptr())),
                (vp(hist.data_ptr()), 4, 16, 0, vp(fl.data_ptr())), (vp(hist.data_ptr()), 4, 16, 500, None)):
        assert L.st_shot_boundaries(h, bad[0], bad[1], bad[2], bad[3], 2.5, bad[4], None) == _native.ST_ERR_INVALID
    assert hip_ctx.shot_boundaries(hist) == []
