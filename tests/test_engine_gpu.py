"""GPU: the reference's own op tests (scannertools/tests/test_all.py:141-177,222-233) replayed
against this build -- Python front-end -> mini engine -> Scanner-style kernel classes -> C ABI ->
HIP -- with the value checks the reference lacks (every row compared with the CPU oracle)."""
import numpy as np
import pytest
import torch

import oracle
from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, NamedVideoStream, PerfParams
from util import texture_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    """Stands in for scannertools_infra.tests.sc: a client with 'test1' ingested."""
    client = Client()
    frames, _ = texture_stream(7, 60, 120, 160)
    client.ingest_frames("test1", frames)
    client.ingest_frames("test1_gpu", torch.from_numpy(frames).cuda())
    return client


class DeviceTestBench:
    def test_cpu(self, sc):
        self.run(sc, DeviceType.CPU)

    def test_gpu(self, sc):
        self.run(sc, DeviceType.GPU)


class TestHistogram(DeviceTestBench):
    def run(self, sc, device):
        input = NamedVideoStream(sc, 'test1')
        frame = sc.io.Input([input])
        hist = sc.ops.Histogram(frame=frame, device=device)
        output = NamedStream(sc, 'test_hist')
        output_op = sc.io.Output(hist, [output])
        sc.run(output_op, PerfParams.estimate(), cache_mode=CacheMode.Overwrite, show_progress=False)
        first = next(output.load())
        assert len(first) == 3 and all(c.dtype == np.int32 and c.shape == (16,) for c in first)
        frames = sc._videos['test1']
        assert output.len() == len(frames)
        for i, h in enumerate(output.load()):
            np.testing.assert_array_equal(np.stack(h), oracle.hist_u8c3(frames[i], 16))


class TestHistogramBatched(DeviceTestBench):
    def run(self, sc, device):
        frame = sc.io.Input([NamedVideoStream(sc, 'test1_gpu' if device == DeviceType.GPU else 'test1')])
        hist = sc.ops.Histogram(frame=frame, device=device, batch=17, bins=256)
        output = NamedStream(sc, 'test_hist_b')
        sc.run(sc.io.Output(hist, [output]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
        frames = sc._videos['test1']
        for i, h in enumerate(output.load()):
            np.testing.assert_array_equal(np.stack(h), oracle.hist_u8c3(frames[i], 256))


class TestOpticalFlow(DeviceTestBench):
    def run(self, sc, device):
        input = NamedVideoStream(sc, 'test1')
        frame = sc.io.Input([input])
        flow = sc.ops.OpticalFlow(frame=frame, stencil=[-1, 0], device=device)
        flow_range = sc.streams.Range(flow, ranges=[{'start': 0, 'end': 50}])
        output = NamedStream(sc, 'test_flow')
        output_op = sc.io.Output(flow_range, [output])
        sc.run(output_op, PerfParams.estimate(), cache_mode=CacheMode.Overwrite, show_progress=False)
        assert output.len() == 50

        flow_array = next(output.load())
        assert flow_array.dtype == np.float32
        assert flow_array.shape[0] == 120
        assert flow_array.shape[1] == 160
        assert flow_array.shape[2] == 2
        frames = sc._videos['test1']
        for i, fl in enumerate(output.load()):
            a, b = frames[max(i - 1, 0)], frames[i]      # stencil [-1, 0]; row 0 clamps to (0, 0)
            ref = oracle.optical_flow_rgb(a, b)
            assert np.abs(fl - ref).max() <= 5e-3
            if i in (0, 1, 25):
                assert np.linalg.norm(fl - ref) <= 1e-4 * max(np.linalg.norm(ref), 1e-30) + 1e-6


class TestOpticalFlowDefaultStencilBatched(DeviceTestBench):
    def run(self, sc, device):
        frame = sc.io.Input([NamedVideoStream(sc, 'test1')])
        flow = sc.ops.OpticalFlow(frame=frame, device=device, batch=8)      # registered stencil {0, 1}
        gathered = sc.streams.Gather(flow, [[3, 4, 5, 40, 59]])
        output = NamedStream(sc, 'test_flow_g')
        sc.run(sc.io.Output(gathered, [output]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
        frames = sc._videos['test1']
        rows = [3, 4, 5, 40, 59]
        assert output.len() == len(rows)
        for r, fl in zip(rows, output.load()):
            ref = oracle.optical_flow_rgb(frames[r], frames[min(r + 1, 59)])   # last row: edge clamp
            assert np.abs(fl - ref).max() <= 5e-3


class TestFlowHistogramPipeline(DeviceTestBench):
    """The legacy flow-histogram graph (old/histograms.py:63-78): OpticalFlow -> FlowHistogram."""

    def run(self, sc, device):
        frame = sc.io.Input([NamedVideoStream(sc, 'test1')])
        flow = sc.ops.OpticalFlow(frame=frame, device=device, batch=8)
        fh = sc.ops.FlowHistogram(flow=flow, device=device, batch=5)
        ranged = sc.streams.Range(fh, ranges=[{'start': 0, 'end': 12}])
        out_h, out_f = NamedStream(sc, 'test_flow_hist'), NamedStream(sc, 'test_flow_hist_src')
        sc.run([sc.io.Output(ranged, [out_h]), sc.io.Output(sc.streams.Range(flow, ranges=[{'start': 0, 'end': 12}]), [out_f])],
               PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
        assert out_h.len() == 12
        first = next(out_h.load())
        assert len(first) == 2 and all(c.dtype == np.int32 and c.shape == (64,) for c in first)
        for h, fl in zip(out_h.load(), out_f.load()):
            np.testing.assert_array_equal(np.stack(h), oracle.flow_hist(fl))
            assert np.stack(h)[1].sum() == 120 * 160


def test_draw_flow_op(sc):
    """sc.ops.DrawFlow (vis.py:8-12) on the output of OpticalFlow."""
    frame = sc.io.Input([NamedVideoStream(sc, 'test1')])
    flow = sc.ops.OpticalFlow(frame=frame, stencil=[-1, 0], device=DeviceType.GPU, batch=8)
    vis = sc.ops.DrawFlow(frame=frame, flow=flow)
    out_v, out_f = NamedStream(sc, 'test_draw'), NamedStream(sc, 'test_draw_src')
    pick = [[0, 1, 7, 30]]
    sc.run([sc.io.Output(sc.streams.Gather(vis, pick), [out_v]), sc.io.Output(sc.streams.Gather(flow, pick), [out_f])],
           PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
    frames = sc._videos['test1']
    assert out_v.len() == 4
    for r, pic, fl in zip(pick[0], out_v.load(), out_f.load()):
        assert pic.dtype == np.uint8 and pic.shape == (120, 320, 3)
        np.testing.assert_array_equal(pic, oracle.draw_flow(frames[r], fl))


def test_shot_detection(sc):
    """tests/test_all.py:222-233 with a stream whose cuts are planted, so the count is known."""
    rng = np.random.default_rng(0)
    n, h, w = 600, 36, 48
    cuts = [97, 230, 231 + 150, 500]
    frames = np.empty((n, h, w, 3), np.uint8)
    base = rng.integers(0, 256, (h, w, 3))
    for i in range(n):
        if i in cuts:
            base = rng.integers(0, 256, (h, w, 3))
        frames[i] = np.clip(base + rng.integers(-3, 4, (h, w, 3)), 0, 255)
    sc.ingest_frames('shots', frames)
    input = NamedVideoStream(sc, 'shots')
    frame = sc.io.Input([input])
    range_frame = sc.streams.Range(frame, [{'start': 0, 'end': 1000}])
    hist = sc.ops.Histogram(frame=range_frame)
    boundaries = sc.ops.ShotBoundaries(histograms=hist)
    output = NamedStream(sc, 'output')
    output_op = sc.io.Output(boundaries, [output])
    sc.run(
        output_op, PerfParams.manual(work_packet_size=1000, io_packet_size=1000, pipeline_instances_per_node=1),
        cache_mode=CacheMode.Overwrite, show_progress=False)
    found = next(output.load(rows=[0]))
    assert found == cuts
    assert all(r is None for r in list(output.load())[1:])
    # identical to the reference algorithm run on oracle histograms
    assert found == oracle.shot_boundaries(np.stack([oracle.hist_u8c3(f, 16) for f in frames]))


def test_engine_frees_every_output_buffer(sc):
    assert sc.live_device_buffers() == 0


def test_cache_mode_error(sc):
    frame = sc.io.Input([NamedVideoStream(sc, 'test1')])
    hist = sc.ops.Histogram(frame=sc.streams.Range(frame, [{'start': 0, 'end': 2}]), device=DeviceType.GPU)
    out = NamedStream(sc, 'test_hist')        # exists from TestHistogram
    with pytest.raises(RuntimeError):
        sc.run(sc.io.Output(hist, [out]), PerfParams.estimate())
    with pytest.raises(KeyError):
        NamedStream(sc, 'never_written').len()


def test_front_end_runners(sc):
    """compute_flow / compute_histograms / compute_hsv_histograms / compute_flow_histograms: the
    graphs of old/optical_flow.py and old/histograms.py, checked row by row."""
    import scannertools_amd.imgproc  # noqa: F401  (loads the op library, like scannertools.imgproc)
    from scannertools_amd.histograms import compute_flow_histograms, compute_histograms, compute_hsv_histograms
    from scannertools_amd.optical_flow import compute_flow
    frames = sc._videos['test1']
    pick = [0, 5, 6, 20, 59]
    (flow,) = compute_flow(sc, ['test1'], frames=[pick], batch=4)
    assert flow.len() == len(pick)
    for i, fl in enumerate(flow.load()):
        ref = oracle.optical_flow_rgb(frames[pick[i]], frames[pick[min(i + 1, len(pick) - 1)]])
        assert fl.shape == (120, 160, 2) and np.abs(fl - ref).max() <= 5e-3
    (hist,) = compute_histograms(sc, ['test1'])
    for i, h in enumerate(hist.load()):
        np.testing.assert_array_equal(np.stack(h), oracle.hist_u8c3(frames[i], 16))
    (hsv,) = compute_hsv_histograms(sc, ['test1'], batch=16)
    for i, h in enumerate(hsv.load()):
        ref = oracle.cvt_color(np.ascontiguousarray(frames[i][..., ::-1]), oracle.COLOR_BGR2HSV)     # RGB2HSV
        np.testing.assert_array_equal(np.stack(h), oracle.hist_u8c3(ref, 16))
    (fh,) = compute_flow_histograms(sc, ['test1'], width=80, height=60)
    assert fh.len() == 60
    small = [oracle.resize_u8(f, 80, 60) for f in frames]
    for i, h in enumerate(fh.load()):
        if i in (0, 17, 59):
            flow_ref = oracle.optical_flow_rgb(small[i], small[min(i + 1, 59)])
            assert abs(int(np.stack(h)[0].sum()) - 80 * 60) <= 0 and np.stack(h)[1].sum() == 80 * 60
            # histogram of the oracle's flow agrees except for vectors that straddle a bin edge
            assert np.abs(np.stack(h) - oracle.flow_hist(flow_ref)).sum() <= 8


def test_device_buffer_pool_is_bounded_and_drains(sc):
    """The engine's device-buffer pool (scanner_shim/shim.cpp): freed blocks are kept for reuse, stay inside the cap
    (SCANNER_SHIM_DEV_POOL_MB, default 4 GB), and go back to the driver on request -- what an allocation that runs out of
    memory does before it gives up."""
    frame = sc.io.Input([NamedVideoStream(sc, 'test1_gpu')])
    out = NamedStream(sc, 'pool_probe')
    sc.run(sc.io.Output(sc.ops.OpticalFlow(frame=frame, device=DeviceType.GPU, batch=8), [out]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
    assert sc.live_device_buffers() == 0
    held = sc.device_pool_bytes()
    assert 0 < held <= 4 << 30
    free0 = torch.cuda.mem_get_info()[0]
    assert sc.drain_device_pool() == held and sc.device_pool_bytes() == 0
    assert torch.cuda.mem_get_info()[0] >= free0 + held // 2          # the driver got the memory back
    sc.run(sc.io.Output(sc.ops.OpticalFlow(frame=frame, device=DeviceType.GPU, batch=8), [out]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
    assert sc.device_pool_bytes() > 0                                  # and the pool fills again


def test_front_end_defaults_select_the_unbatched_path_1080p():
    """The front-ends pass what the reference's passes (old/optical_flow.py:19-23: no ``batch=``; old/histograms.py:10-15:
    ``batch=1``), so an unchanged graph drives ONE pair / ONE frame per ``execute()``: that path at 1080p, every row against
    the oracle; and the same rows again with the speed switch (``batch=``), bit for bit."""
    import scannertools_amd.imgproc  # noqa: F401
    from scannertools_amd.histograms import compute_histograms
    from scannertools_amd.optical_flow import build_pipeline, compute_flow
    from util import assert_flow_close
    client = Client()
    frames, _ = texture_stream(11, 3, 1080, 1920)
    client.ingest_frames("hd", torch.from_numpy(frames).cuda())
    node = build_pipeline(client, client.io.Input([NamedVideoStream(client, "hd")]))['flow']
    assert node.batch == 1                      # what Scanner does with an op that got no batch=
    (flow,) = compute_flow(client, ["hd"])
    rows = list(flow.load())
    assert len(rows) == 3
    for i, fl in enumerate(rows):
        j = min(i + 1, 2)
        assert_flow_close(fl, oracle.optical_flow_rgb(frames[i], frames[j]), frames[i], frames[j], "default compute_flow row %d" % i)
    (flow_b,) = compute_flow(client, ["hd"], batch=3, suffix="flow_b")
    for a, b in zip(rows, flow_b.load()):
        np.testing.assert_array_equal(a, b)
    (hist,) = compute_histograms(client, ["hd"])
    for i, h in enumerate(hist.load()):
        np.testing.assert_array_equal(np.stack(h), oracle.hist_u8c3(frames[i], 16))


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["own", "torchrun"])
def test_bench_two_ranks_sharing_this_gpu(launcher):
    """The multi-rank path of bench.py on real hardware with the one GPU a test box has: two rank processes (started by
    bench.py itself, and by torch.distributed.run as the driver does) share device 0 and synchronise over gloo
    (ST_BENCH_SHARE_GPU=1; RCCL refuses two ranks on one device).  Per-rank shards, barriers, max-over-ranks time and
    rank 0's single JSON line are the code the 8-GPU run uses."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["ST_BENCH_SHARE_GPU"] = "1"
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--height", "270", "--width", "480"]
    cmd = [sys.executable] + tail if launcher == "own" else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                                              "--master-addr", "127.0.0.1", "--master-port", "29581"] + tail
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_share_gpus"] is True and d["value"] > 0 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 8 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]   # whole-job aggregate


@pytest.mark.gpu
def test_shot_pipeline_two_ranks_sharing_this_gpu():
    """Config 3's multi-rank path on real hardware (scripts/shot_pipeline.py --gpus 2, ranks sharing device 0, gloo): shards of a
    stream with planted cuts -> Histogram kernel per rank -> gather on rank 0 -> ShotBoundaries on the device, which the run itself
    compares with the host op."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["ST_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "shot_pipeline.py"), "--gpus", "2", "--frames", "3000", "--height", "72", "--width", "128",
                        "--cuts", "4", "--chunk", "500"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["frames"] == 3000 and d["planted_found"] is True and d["shot_boundaries_s"] < d["shot_boundaries_host_s"]
