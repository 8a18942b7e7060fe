"""Pose path (BASELINE config 5): the two deterministic ends of the CPM2 /
OpenPose COCO-18 pipeline -- CPM2Input (frame -> network input) and CPM2Output (heat maps + joint
candidates -> people) -- against the oracle's restatements of
scannertools_caffe_cpp/cpm2_input_kernel_gpu.cpp:104-140 and cpm2_output_kernel_cpu.cpp:362-689.

CPU tests: the oracle against planted people, the host-registered CPM2Output kernel class against the
oracle (byte-level output format included).  GPU tests: CPM2Input bit-exact (float results are exact
multiples of 1/256), the limb-scoring kernel bit-exact, the GPU-registered CPM2Output against the host one."""
import struct

import numpy as np
import pytest

import oracle
from scannertools_amd import types as st_types
from util import random_frames, synthetic_pose_maps


# ------------------------------------------------------------------------------------------- CPU
def test_cpm2_geometry_rules():
    # cpm2_input_kernel_gpu.cpp:48-55: truncating float product, padding up to a multiple of 8
    assert oracle.cpm2_geometry(1080, 1920, 368 / 1080.) == (368, 654, 368, 656)
    assert oracle.cpm2_geometry(480, 640, 0.5) == (240, 320, 240, 320)
    assert oracle.cpm2_geometry(100, 100, 0.33) == (33, 33, 40, 40)
    assert oracle.cpm2_geometry(9, 9, 1.0) == (9, 9, 16, 16)
    with pytest.raises(ValueError):
        oracle.cpm2_geometry(10, 10, 0.01)
    from scannertools_amd.hip import cpm2_geometry
    for h, w, s in ((1080, 1920, 368 / 1080.), (480, 640, 0.5), (100, 100, 0.33), (9, 9, 1.0), (2160, 3840, 0.17037)):
        assert cpm2_geometry(h, w, s) == oracle.cpm2_geometry(h, w, s)


def test_cpm2_input_oracle_known_answers():
    f = random_frames(3, 1, 40, 56)[0]
    out = oracle.cpm2_input(f, 1.0)                       # scale 1: no resampling, no padding (40, 56 are multiples of 8)
    assert out.shape == (3, 40, 56)
    np.testing.assert_array_equal(out, (f[..., ::-1].transpose(2, 0, 1).astype(np.float32) / 256 - 0.5))   # planes B, G, R
    out = oracle.cpm2_input(f[:37, :50].copy(), 1.0)      # 37 x 50 -> padded to 40 x 56 with 128 -> 0.0
    assert out.shape == (3, 40, 56) and (out[:, 37:, :] == 0).all() and (out[:, :, 50:] == 0).all()
    const = np.full((90, 120, 3), 77, np.uint8)
    assert (oracle.cpm2_input(const, 0.4)[:, :36, :48] == np.float32(77 / 256 - 0.5)).all()
    half = oracle.cpm2_input(f, 0.5)                      # the resize is cv::resize's bicubic on the swapped frame
    np.testing.assert_array_equal(half[:, :20, :28] * 256 + 128,
                                  oracle.resize_u8(f[..., ::-1].copy(), 28, 20, oracle.INTER_CUBIC).transpose(2, 0, 1))


def test_connect_limbs_recovers_planted_people():
    H, W = 184, 328
    for seed, n in ((1, 1), (2, 3), (3, 5)):
        # (no false candidates here: one that happens to lie on a limb scores like the true joint, the
        # matching being by affinity alone; the comparisons with the oracle below keep the clutter)
        hm, peaks, truth = synthetic_pose_maps(seed, H, W, n, drop=0.0, clutter=0)
        people = oracle.cpm2_connect_limbs_coco(hm, peaks, frame_h=H * 3, frame_w=W * 3)
        assert len(people) == n
        for p in range(n):
            # every planted person comes back whole: match by the neck, compare all joints
            d = [np.abs(q[1, :2] / 3 - np.round(truth[p, 1])).sum() for q in people]
            q = people[int(np.argmin(d))]
            assert min(d) == 0
            np.testing.assert_allclose(q[:, :2] / 3, np.round(truth[p]), atol=1e-4)
            assert (q[:, 2] >= 0.5).all()
    # no candidates at all -> nobody; a lone part is dropped by the 3-joint minimum
    hm, peaks, _ = synthetic_pose_maps(4, H, W, 0, clutter=0)
    assert len(oracle.cpm2_connect_limbs_coco(hm, peaks, H, W)) == 0
    peaks[5, 0, 0] = 1
    peaks[5, 1] = (10, 10, 0.9)
    assert len(oracle.cpm2_connect_limbs_coco(hm, peaks, H, W)) == 0


def _frame_info_bytes(h, w, c=3, ftype=0):
    return struct.pack("<4i", h, w, c, ftype)


def _run_cpm2_output(device, maps, peaks, frame_h, frame_w, scale, batch=2):
    from scannertools_amd.engine import Client, NamedStream, PerfParams, CacheMode, _InputNode

    class _Rows:  # a column of ready-made rows
        def __init__(self, rows):
            self.rows_ = rows

        def length(self):
            return len(self.rows_)

        def rows(self, idx):
            return [self.rows_[i] for i in idx]

    sc = Client()
    info = _Rows([_frame_info_bytes(frame_h, frame_w)] * len(maps))
    node = sc.ops.CPM2Output(cpm2_resized_map=_Rows(list(maps)), cpm2_joints=_Rows(list(peaks)), original_frame_info=info,
                             scale=scale, device=device, batch=batch)
    out = NamedStream(sc, "poses")
    sc.run(sc.io.Output(node, [out]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
    raw = sc._tables["poses"][0]
    return raw, list(out.load())


def _pose_case(seed, n_people, frame_h=540, frame_w=960, scale=184 / 540.):
    _, _, H, W = oracle.cpm2_geometry(frame_h, frame_w, scale)
    hm, peaks, _ = synthetic_pose_maps(seed, H, W, n_people)
    return hm, peaks, (frame_h, frame_w, scale)


def test_cpm2_output_host_kernel_matches_oracle():
    """DeviceType::CPU registration (the reference's): Python engine -> kernel class -> cpm2_parse.h."""
    from scannertools_amd.engine import DeviceType
    cases = [_pose_case(s, n) for s, n in ((11, 2), (12, 4), (13, 0), (14, 7))]
    fh, fw, scale = cases[0][2]
    raw, got = _run_cpm2_output(DeviceType.CPU, [c[0] for c in cases], [c[1] for c in cases], fh, fw, scale)
    for (hm, peaks, _), g, r in zip(cases, got, raw):
        ref = oracle.cpm2_connect_limbs_coco(hm, peaks, fh, fw)
        assert g.shape == ref.shape
        np.testing.assert_array_equal(g, ref)
        # byte format: u64 people, per person u64 18, per joint i32 size + proto3 Point
        assert struct.unpack_from("<Q", r, 0)[0] == len(ref)
        if len(ref):
            assert struct.unpack_from("<Q", r, 8)[0] == 18
    assert sum(len(g) for g in got) >= 10


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("h,w,scale", [(1080, 1920, 368 / 1080.), (480, 640, 0.5), (97, 131, 0.77), (64, 64, 1.0),
                                        (37, 50, 1.0), (240, 320, 1.7), (2160, 3840, 368 / 2160.)])
def test_cpm2_input_bit_exact(hip_ctx, h, w, scale):
    import torch
    frames = random_frames(h + w, 2, h, w)
    got = hip_ctx.cpm2_input(torch.from_numpy(frames).cuda(), scale).cpu().numpy()
    for i in range(2):
        np.testing.assert_array_equal(got[i], oracle.cpm2_input(frames[i], scale))


@pytest.mark.gpu
def test_cpm2_limb_scores_bit_exact(hip_ctx):
    import torch
    cases = [_pose_case(s, n) for s, n in ((21, 3), (22, 6), (23, 0))]
    hm = torch.from_numpy(np.stack([c[0] for c in cases])).cuda()
    pk = torch.from_numpy(np.stack([c[1] for c in cases])).cuda()
    got = hip_ctx.cpm2_limb_scores(hm, pk).cpu().numpy()
    for i, (m, p, _) in enumerate(cases):
        np.testing.assert_array_equal(got[i], oracle.cpm2_limb_scores(m, p))
    assert (got[0] >= 0).sum() > 20 and (got[2] >= 0).sum() == 0


@pytest.mark.gpu
def test_cpm2_ops_through_the_kernel_classes():
    """CPM2Input through both registrations; CPM2Output on device columns == on host columns == oracle."""
    import torch
    from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, NamedVideoStream, PerfParams
    frames = random_frames(5, 5, 270, 480)
    scale = 184 / 270.
    sc = Client()
    sc.ingest_frames("v", frames)
    frame = sc.io.Input([NamedVideoStream(sc, "v")])
    for device in (DeviceType.GPU, DeviceType.CPU):
        out = NamedStream(sc, "cpm2_in")
        sc.run(sc.io.Output(sc.ops.CPM2Input(frame=frame, scale=scale, device=device, batch=3), [out]),
               PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
        for i, o in enumerate(out.load()):
            np.testing.assert_array_equal(o, oracle.cpm2_input(frames[i], scale))
    cases = [_pose_case(s, n) for s, n in ((31, 2), (32, 5), (33, 1))]
    fh, fw, sc_ = cases[0][2]
    raw_h, host = _run_cpm2_output(DeviceType.CPU, [c[0] for c in cases], [c[1] for c in cases], fh, fw, sc_)
    raw_d, dev = _run_cpm2_output(DeviceType.GPU, [torch.from_numpy(c[0]).cuda() for c in cases],
                                  [torch.from_numpy(c[1]).cuda() for c in cases], fh, fw, sc_, batch=3)
    assert raw_h == raw_d
    for (hm, peaks, _), g in zip(cases, dev):
        np.testing.assert_array_equal(g, oracle.cpm2_connect_limbs_coco(hm, peaks, fh, fw))
