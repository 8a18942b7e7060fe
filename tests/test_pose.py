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
def test_cpm2_scale_for_a_given_height():
    """368.f / 1080 resizes to 367 rows under the truncating size rule; the OpenPose op asks for the scale that lands
    on exactly its network input height (host-only entry point, no GPU)."""
    from scannertools_amd.hip import cpm2_geometry, cpm2_scale_for_height
    for h in (1080, 2160, 720, 480, 96, 1, 367, 369, 4320):
        for target in (368, 313, 184, 8):
            s = cpm2_scale_for_height(h, target)
            assert cpm2_geometry(h, 16 * h, s)[0] == target
            assert abs(s - target / h) <= 4e-7 * max(1.0, target / h)
    naive = float(np.float32(368) / np.float32(1080))
    assert cpm2_geometry(1080, 1920, naive)[0] in (367, 368) and cpm2_geometry(1080, 1920, cpm2_scale_for_height(1080, 368))[0] == 368
    with pytest.raises(Exception):
        cpm2_scale_for_height(0, 368)


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


# ---- the `resize` and `nms` layers between the network and CPM2Output -----------------------------------------
def test_resize_maps_oracle_known_answers():
    """Properties the interpolation formula implies: a constant map stays constant (the cubic's weights sum to 1),
    an affine ramp is reproduced in the interior (Catmull-Rom is exact on polynomials of degree <= 1 ... 3), and the
    8x geometry puts output pixel 8k + 3.5 on source pixel k."""
    const = np.full((2, 6, 9), 0.37, np.float32)
    np.testing.assert_allclose(oracle.cpm2_resize_maps(const, 48, 72), 0.37, rtol=0, atol=1e-6)
    yy, xx = np.mgrid[0:12, 0:15].astype(np.float32)
    ramp = (0.25 * xx + 0.5 * yy)[None]
    up = oracle.cpm2_resize_maps(ramp, 96, 120)
    y, x = np.mgrid[0:96, 0:120]
    expect = 0.25 * ((x - 3.5) / 8.) + 0.5 * ((y - 3.5) / 8.)
    inner = (slice(None), slice(16, 80), slice(16, 104))
    np.testing.assert_allclose(up[inner], expect[None][inner], atol=2e-5)
    assert up.dtype == np.float32 and up.shape == (1, 96, 120)
    # non-integer ratio: still a convex-ish interpolation of a bounded map (Catmull-Rom overshoot <= 12.5 % per axis)
    rng = np.random.default_rng(3)
    m = rng.random((1, 7, 5), dtype=np.float32)
    r = oracle.cpm2_resize_maps(m, 23, 31)
    assert r.min() > -0.3 and r.max() < 1.3


def test_nms_oracle_known_answers():
    m = np.zeros((2, 9, 11), np.float32)
    m[0, 3, 4] = 0.9          # a peak
    m[0, 3, 8] = 0.04         # below the threshold
    m[0, 0, 5] = 0.8          # on the border: never a peak
    m[0, 6, 2] = m[0, 6, 3] = 0.7   # plateau: neither exceeds the other
    m[0, 7, 9] = 0.6
    m[1, 1, 1] = 0.3
    j = oracle.cpm2_nms(m, parts=2, max_peaks=4, threshold=0.05)
    assert j.shape == (2, 5, 3)
    np.testing.assert_array_equal(j[0, :3], np.array([[2, 0, 0], [4, 3, 0.9], [9, 7, 0.6]], np.float32))
    np.testing.assert_array_equal(j[0, 3:], 0)
    np.testing.assert_array_equal(j[1, :2], np.array([[1, 0, 0], [1, 1, 0.3]], np.float32))
    # more peaks than max_peaks: the first ones in raster order, count capped
    many = np.zeros((1, 20, 20), np.float32)
    many[0, 1:19:2, 1:19:2] = np.linspace(0.1, 0.9, 81).reshape(9, 9)
    j = oracle.cpm2_nms(many, parts=1, max_peaks=5, threshold=0.05)
    assert j[0, 0, 0] == 5
    np.testing.assert_array_equal(j[0, 1:, :2], [[1, 1], [3, 1], [5, 1], [7, 1], [9, 1]])


def test_caffemodel_reader_round_trip(tmp_path):
    """read_caffemodel on a file written with the wire-format helpers: current (`layer` = 100, shape message) and V1
    (`layers` = 2, num/channels/height/width) layouts; layer names of the published prototxt."""
    from scannertools_amd import _proto, pose_net
    rng = np.random.default_rng(0)
    w = rng.standard_normal((4, 3, 3, 3)).astype(np.float32)
    b = rng.standard_normal(4).astype(np.float32)

    def blob_new(a):
        return _proto.message(7, _proto.message(1, b"".join(_proto._varint(d) for d in a.shape))) + _proto.message(5, a.astype("<f4").tobytes())

    def blob_v1(a):
        dims = (list(a.shape) + [1, 1, 1])[:4] if a.ndim > 1 else [1, 1, 1, a.shape[0]]
        return b"".join(_proto._varint(i + 1 << 3) + _proto._varint(d) for i, d in enumerate(dims)) + _proto.message(5, a.astype("<f4").tobytes())

    new = _proto.message(1, b"net") + _proto.message(100, _proto.message(1, b"conv1_1") + _proto.message(2, b"Convolution") +
                                                     _proto.message(7, blob_new(w)) + _proto.message(7, blob_new(b)))
    old = _proto.message(2, _proto.message(4, b"conv1_1") + _proto.message(6, blob_v1(w)) + _proto.message(6, blob_v1(b)))
    for name, buf in (("new", new), ("old", old)):
        p = tmp_path / (name + ".caffemodel")
        p.write_bytes(buf)
        got = pose_net.read_caffemodel(str(p))
        assert list(got) == ["conv1_1"]
        np.testing.assert_array_equal(got["conv1_1"][0].reshape(w.shape), w)
        np.testing.assert_array_equal(got["conv1_1"][1].reshape(-1), b)
    with pytest.raises(ValueError):
        list(_proto.fields(new[:-3]))   # truncated file
    names = pose_net.caffe_layer_names()
    assert len(names) == len(pose_net.all_layers()) == 92 and len(set(names)) == 92
    assert names[12] == "conv5_1_CPM_L1" and names[-1] == "Mconv7_stage6_L2" and names[22] == "Mconv1_stage2_L1"


def _write_model(path, layers, corrupt=None):
    """A caffemodel with zero weights of the architecture's sizes for the given (caffe name, cin, cout, k) layers."""
    from scannertools_amd import _proto
    with open(path, "wb") as fh:
        fh.write(_proto.message(1, b"pose"))
        for name, ci, co, k in layers:
            wn = co * ci * k * k - (1 if corrupt == name else 0)
            blobs = b"".join(_proto.message(7, _proto.message(5, np.zeros(n, "<f4").tobytes())) for n in (wn, co))
            fh.write(_proto.message(100, _proto.message(1, name.encode()) + _proto.message(2, b"Convolution") + blobs))


def test_op_library_checks_a_model_file(tmp_path):
    """scannertools_caffe_check_model (the C++ reader CPM2KernelHIP uses, no GPU needed): a complete file passes with 92
    layers; a missing layer, a wrong blob size, a truncated file and a file that is not a NetParameter are refused
    with a message that says what is wrong."""
    from scannertools_amd import pose_net
    layers = [(cn, ci, co, k) for (_, ci, co, k, _), cn in zip(pose_net.all_layers(), pose_net.caffe_layer_names())]
    good = tmp_path / "good.caffemodel"
    _write_model(good, layers)
    assert pose_net.check_caffemodel(good) == 92
    missing = tmp_path / "missing.caffemodel"
    _write_model(missing, [l for l in layers if l[0] != "Mconv3_stage4_L2"])
    with pytest.raises(ValueError, match="Mconv3_stage4_L2"):
        pose_net.check_caffemodel(missing)
    wrong = tmp_path / "wrong.caffemodel"
    _write_model(wrong, layers, corrupt="conv4_3_CPM")
    with pytest.raises(ValueError, match="conv4_3_CPM"):
        pose_net.check_caffemodel(wrong)
    cut = tmp_path / "cut.caffemodel"
    cut.write_bytes(good.read_bytes()[:1000003])
    with pytest.raises(ValueError, match="NetParameter|no weights"):
        pose_net.check_caffemodel(cut)
    junk = tmp_path / "junk.caffemodel"
    junk.write_bytes(np.random.default_rng(0).integers(0, 256, 4096, dtype=np.uint8).tobytes())
    with pytest.raises(ValueError):
        pose_net.check_caffemodel(junk)
    with pytest.raises(ValueError, match="cannot read"):
        pose_net.check_caffemodel(tmp_path / "absent.caffemodel")
    with pytest.raises(ValueError, match="cannot read"):
        pose_net.check_caffemodel(tmp_path)  # a directory: fopen() succeeds on it
    hostile = tmp_path / "hostile.caffemodel"
    hostile.write_bytes(b"\xa2\x06" + b"\xff" * 9 + b"\x01\x00")  # length-delimited field with a length of 2**64 - 1
    with pytest.raises(ValueError, match="NetParameter"):
        pose_net.check_caffemodel(hostile)


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w,C,th,tw,chan", [(2, 6, 9, 57, 48, 72, None), (1, 46, 82, 192, 368, 656, "pose"), (3, 7, 5, 8, 23, 31, [7, 0, 3]),
                                                (1, 5, 5, 4, 5, 5, None), (1, 12, 16, 3, 6, 8, None)])
def test_resize_maps_bit_exact(hip_ctx, n, h, w, C, th, tw, chan):
    import torch
    from scannertools_amd import pose_net
    rng = np.random.default_rng(h * w + C)
    maps = rng.standard_normal((n, h, w, C)).astype(np.float32)
    if chan == "pose":
        chan = [pose_net.OFF_HEAT + i for i in range(19)] + [pose_net.OFF_PAF + i for i in range(38)]
    got = hip_ctx.cpm2_resize_maps(torch.from_numpy(maps).cuda(), th, tw, chan_map=chan).cpu().numpy()
    sel = chan if chan is not None else list(range(C))
    for i in range(n):
        np.testing.assert_array_equal(got[i], oracle.cpm2_resize_maps(np.ascontiguousarray(maps[i].transpose(2, 0, 1)[sel]), th, tw))


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,max_peaks,density", [(368, 656, 64, 0.0005), (368, 656, 64, 0.02), (37, 53, 8, 0.05), (3, 3, 4, 1.0), (2, 9, 4, 0.5),
                                                  (64, 1030, 1024, 0.2)])
def test_nms_bit_exact(hip_ctx, h, w, max_peaks, density):
    """Sparse and dense peak populations (more than max_peaks in one 1024-pixel round, peaks on round boundaries,
    plateaus, borders), rows of other lengths than the workgroup's round."""
    import torch
    rng = np.random.default_rng(h + w)
    maps = (rng.random((2, 20, h, w)) * 0.04).astype(np.float32)
    mask = rng.random(maps.shape) < density
    maps[mask] = rng.random(int(mask.sum())).astype(np.float32)
    maps[:, :, ::7, ::5] = np.float32(0.5)   # equal values: plateaus wherever two of them touch
    got = hip_ctx.cpm2_nms(torch.from_numpy(maps).cuda(), parts=18, max_peaks=max_peaks, threshold=0.05).cpu().numpy()
    for i in range(2):
        np.testing.assert_array_equal(got[i], oracle.cpm2_nms(maps[i], 18, max_peaks, 0.05))
    if h > 2 and density < 1:
        assert got[:, :, 0, 0].max() > 0


@pytest.mark.gpu
def test_resize_nms_output_chain_recovers_planted_people(hip_ctx):
    """Low-resolution maps with planted people -> resize x8 -> nms -> CPM2Output (limb scores on the GPU, assembly
    on the host): everybody is found, and the chain equals the oracle's chain on the same low-resolution maps."""
    import torch
    H8, W8, people = 46, 82, 3
    hm_lo, _, truth = synthetic_pose_maps(5, H8, W8, people, clutter=0, drop=0.0)
    # joints become blobs in the part planes (the planted candidates of synthetic_pose_maps are not used)
    hm_lo[:19] = 0
    yy, xx = np.mgrid[0:H8, 0:W8]
    for p in range(people):
        for j in range(18):
            hm_lo[j] += np.exp(-((xx - truth[p, j, 0]) ** 2 + (yy - truth[p, j, 1]) ** 2) / 1.5).astype(np.float32)
    lo = torch.from_numpy(np.ascontiguousarray(hm_lo.transpose(1, 2, 0))[None]).cuda()
    maps = hip_ctx.cpm2_resize_maps(lo, 8 * H8, 8 * W8)
    joints = hip_ctx.cpm2_nms(maps, parts=18, max_peaks=64, threshold=0.05)
    ref_maps = oracle.cpm2_resize_maps(hm_lo, 8 * H8, 8 * W8)
    ref_joints = oracle.cpm2_nms(ref_maps, 18, 64, 0.05)
    np.testing.assert_array_equal(maps[0].cpu().numpy(), ref_maps)
    np.testing.assert_array_equal(joints[0].cpu().numpy(), ref_joints)
    scores = hip_ctx.cpm2_limb_scores(maps, joints).cpu().numpy()
    got = oracle.cpm2_connect_limbs_coco(ref_maps, ref_joints, 8 * H8, 8 * W8, scores=scores[0])
    assert got.shape[0] == people
    for p in range(people):
        d = np.abs(got[:, 1, :2] - (truth[p, 1] * 8 + 3.5)).sum(axis=1)
        person = got[int(d.argmin())]
        assert np.abs(person[:, :2] - (truth[p] * 8 + 3.5)).max() < 8.0 and (person[:, 2] > 0).all()


def test_resize_merge_oracle_and_pose_type():
    """CPU: merging scales in the oracle (one scale = the plain resize; constants stay constant; a smaller scale sampled
    with its effective extent shows the same picture), and the Pose element type / PoseList reader."""
    from scannertools_amd import pose_detection as pd
    rng = np.random.default_rng(2)
    m = rng.random((2, 6, 9), dtype=np.float32)
    np.testing.assert_array_equal(oracle.cpm2_resize_merge_maps([m], [(6, 9)], 48, 72), oracle.cpm2_resize_maps(m, 48, 72))
    c0, c1 = np.full((1, 6, 9), 0.25, np.float32), np.full((1, 5, 7), 0.75, np.float32)
    np.testing.assert_allclose(oracle.cpm2_resize_merge_maps([c0, c1], [(6, 9), (4.2, 6.3)], 48, 72), 0.5, atol=1e-6)
    yy, xx = np.mgrid[0:12, 0:16].astype(np.float32)
    big = (0.1 * xx + 0.05 * yy)[None]
    # the same ramp seen at 3/4 of the resolution: pixel i of the small map sits at (i + 0.5) / 0.75 - 0.5 of the big one
    small = (0.1 * ((xx[:9, :12] + 0.5) / 0.75 - 0.5) + 0.05 * ((yy[:9, :12] + 0.5) / 0.75 - 0.5))[None].astype(np.float32)
    a = oracle.cpm2_resize_maps(big, 96, 128)
    b = oracle.cpm2_resize_maps(small, 96, 128, eff=(9.0, 12.0))
    np.testing.assert_allclose(a[:, 16:60, 16:80], b[:, 16:60, 16:80], atol=2e-5)

    kp = rng.random((pd.Pose.total_keypoints(), 3)).astype(np.float32)
    p = pd.Pose(0.7, kp)
    q = pd.Pose.deserialize(p.serialize())
    assert abs(q.score() - 0.7) < 1e-6 and np.array_equal(q.pose_keypoints(), kp[:18]) and q.face_keypoints().shape == (70, 3)
    assert [h.shape for h in q.hand_keypoints()] == [(21, 3), (21, 3)]
    assert pd.Pose.kp_size() == 1 + (18 + 70 + 42) * 3 and (pd.Pose.Nose, pd.Pose.Neck, pd.Pose.LEar, pd.Pose.Background) == (0, 1, 17, 18)
    assert pd.pose_list(b"\0\0\0\0") == [] and len(pd.pose_list(p.serialize() * 3)) == 3 and pd.pose_list(None) == []
    (x0, y0), (x1, y1), s = p.body_bbox()
    assert x0 <= x1 and y0 <= y1 and 0 < s < 1
    assert p.distance_to(q) == 0.0 and p.face_bbox()[2] > 0
    img = np.zeros((64, 96, 3), np.uint8)
    assert p.draw(img, thickness=3).any()
    far = pd.Pose(0.1, np.zeros_like(kp))
    assert p.distance_to(far) == float("inf") and far.face_bbox() == [(0, 0), (0, 0), 0]


@pytest.mark.gpu
def test_pose_entry_points_reject_bad_arguments_and_accept_empty_batches(hip_ctx):
    """Error behaviour of the pose entry points of the C ABI: empty batches are no-ops, malformed requests return
    ST_ERR_INVALID / ST_ERR_UNSUPPORTED with a message and leave the context usable."""
    import ctypes
    import torch
    from scannertools_amd import _native
    from scannertools_amd._native import StError
    L, h = hip_ctx._L, hip_ctx._h
    hip_ctx._bind()
    f32 = torch.float32
    # empty batches
    assert tuple(hip_ctx.cpm2_resize_maps(torch.empty((0, 4, 4, 8), dtype=f32, device="cuda"), 8, 8).shape) == (0, 8, 8, 8)
    assert tuple(hip_ctx.cpm2_nms(torch.empty((0, 18, 8, 8), dtype=f32, device="cuda")).shape) == (0, 18, 65, 3)
    assert tuple(hip_ctx.cpm2_limb_scores(torch.empty((0, 57, 8, 8), dtype=f32, device="cuda"), torch.empty((0, 18, 65, 3), dtype=f32, device="cuda")).shape) == (0, 19, 64, 64)
    assert tuple(hip_ctx.cpm2_input(torch.empty((0, 16, 16, 3), dtype=torch.uint8, device="cuda"), 1.0).shape)[0] == 0
    maps = torch.zeros((1, 4, 4, 8), dtype=f32, device="cuda")
    # resize: channel outside the pixel, more than 64 maps, no scales, too many scales
    with pytest.raises(StError, match="channel"):
        hip_ctx.cpm2_resize_maps(maps, 8, 8, chan_map=[0, 9])
    with pytest.raises(StError):
        hip_ctx.cpm2_resize_maps(torch.zeros((1, 2, 2, 80), dtype=f32, device="cuda"), 4, 4)
    with pytest.raises(StError):
        hip_ctx.cpm2_resize_merge_maps([maps] * 9, [(4, 4)] * 9, 8, 8)
    with pytest.raises(StError, match="empty extent"):
        hip_ctx.cpm2_resize_merge_maps([maps], [(0.0, 4.0)], 8, 8)
    # nms: max_peaks out of range
    with pytest.raises(StError):
        hip_ctx.cpm2_nms(torch.zeros((1, 18, 8, 8), dtype=f32, device="cuda"), max_peaks=0)
    with pytest.raises(StError):
        hip_ctx.cpm2_nms(torch.zeros((1, 18, 8, 8), dtype=f32, device="cuda"), max_peaks=5000)
    # conv: channel count not a multiple of 16, even kernel, slice beyond the buffer, misaligned channel offset
    x = torch.zeros((1, 4, 4, 32), dtype=f32, device="cuda")
    w = torch.zeros((64, 3, 3, 32), dtype=f32, device="cuda")
    b = torch.zeros((64,), dtype=f32, device="cuda")
    y = torch.zeros((1, 4, 4, 64), dtype=f32, device="cuda")
    vp = ctypes.c_void_p

    def conv(cin=32, xs=32, xoff=0, k=3, cout=64, cop=64, ys=64, yoff=0):
        return L.st_conv2d_nhwc_f32(h, vp(x.data_ptr()), 1, 4, 4, cin, xs, xoff, vp(w.data_ptr()), vp(b.data_ptr()), k, k, cout, cop, 1, vp(y.data_ptr()), ys, yoff)

    assert conv() == 0
    assert conv(cin=24) == _native.ST_ERR_INVALID and b"multiple of 16" in L.st_ctx_last_error(h)
    assert conv(k=4) == _native.ST_ERR_UNSUPPORTED and conv(k=9) == _native.ST_ERR_UNSUPPORTED
    assert conv(xoff=16) != 0            # 16 + 32 channels > 32
    assert conv(cin=16, xoff=2) != 0     # offset not a multiple of 4
    assert conv(cop=96) != 0 and conv(cout=80) != 0 and conv(yoff=8) != 0
    # the split-bf16 entry points refuse the same things (and a misaligned / missing weight buffer)
    w3 = torch.zeros((w.numel() * 6,), dtype=torch.uint8, device="cuda")
    assert L.st_conv_pack_weights_bf16x3(h, vp(w.data_ptr()), 64, 3, 3, 32, vp(w3.data_ptr())) == 0
    assert L.st_conv_pack_weights_bf16x3(h, vp(w.data_ptr()), 64, 3, 3, 24, vp(w3.data_ptr())) == _native.ST_ERR_INVALID
    assert L.st_conv_pack_weights_bf16x3(h, vp(w.data_ptr()), 64, 3, 3, 32, vp(w3.data_ptr() + 2)) == _native.ST_ERR_INVALID
    assert L.st_conv_pack_weights_bf16x3(h, None, 64, 3, 3, 32, vp(w3.data_ptr())) == _native.ST_ERR_INVALID
    # the size-checked entry point: a buffer sized by the 6-bytes-per-weight rule is too small for a layer that also gets the
    # spatial-tile kernel's copy (3x3 / 7x7 with 128-channel output blocks: 12 bytes per weight) -- refused, nothing written
    need64, need128 = L.st_conv_bf16x3_packed_bytes(64, 3, 3, 32), L.st_conv_bf16x3_packed_bytes(128, 3, 3, 32)
    assert need64 == 64 * 9 * 32 * 6 and need128 == 128 * 9 * 32 * 12
    w128 = torch.randn((128, 3, 3, 32), device="cuda")
    big = torch.full((need128 + 64,), 0xAB, dtype=torch.uint8, device="cuda")
    assert L.st_conv_pack_weights_bf16x3_n(h, vp(w128.data_ptr()), 128, 3, 3, 32, vp(big.data_ptr()), 128 * 9 * 32 * 6) == _native.ST_ERR_INVALID
    assert b"smaller" in L.st_ctx_last_error(h) and bool((big == 0xAB).all())
    assert L.st_conv_pack_weights_bf16x3_n(h, vp(w128.data_ptr()), 128, 3, 3, 32, vp(big.data_ptr()), need128) == 0
    torch.cuda.synchronize()
    assert bool((big[need128:] == 0xAB).all()) and not bool((big[:need128] == 0xAB).all())

    def conv3(cin=32, xs=32, xoff=0, k=3, cout=64, cop=64, ys=64, yoff=0, wp=None):
        return L.st_conv2d_nhwc_bf16x3(h, vp(x.data_ptr()), 1, 4, 4, cin, xs, xoff, vp(w3.data_ptr()) if wp is None else wp, vp(b.data_ptr()), k, k, cout, cop, 1,
                                       vp(y.data_ptr()), ys, yoff)

    assert conv3() == 0
    assert conv3(cin=24) == _native.ST_ERR_INVALID and conv3(k=4) == _native.ST_ERR_UNSUPPORTED
    assert conv3(xoff=16) != 0 and conv3(cop=96) != 0 and conv3(cout=80) != 0 and conv3(yoff=8) != 0
    assert conv3(wp=vp(w3.data_ptr() + 4)) == _native.ST_ERR_INVALID and conv3(wp=None if False else vp(0)) == _native.ST_ERR_INVALID
    assert L.st_maxpool2_nhwc_f32(h, vp(x.data_ptr()), 1, 1, 4, 32, 32, vp(y.data_ptr()), 32) != 0
    assert L.st_maxpool2_nhwc_f32(h, vp(x.data_ptr() + 4), 1, 4, 4, 32, 32, vp(y.data_ptr()), 32) != 0
    assert L.st_planar_to_nhwc_f32(h, vp(x.data_ptr()), 1, 3, 4, 4, vp(y.data_ptr()), 2) != 0
    assert b"planar_to_nhwc" in L.st_ctx_last_error(h)
    # the context still works
    got = hip_ctx.cpm2_resize_maps(torch.ones((1, 4, 4, 8), dtype=f32, device="cuda"), 8, 8)
    assert float((got - 1).abs().max()) < 1e-6


def test_prototxt_is_read_and_checked(tmp_path):
    """The deploy description: the Python reader walks the blobs (channel counts through Concat, ReLU attribution, pooling
    positions) and the op library's reader lists the convolutions; both accept the published architecture, both take the
    layer NAMES from the file (a renamed model with the same structure is usable, its weights found under the new names in
    a caffemodel whose layers are stored in another order, between blob-less layers), both name the first difference of a
    description of another network."""
    from scannertools_amd import pose_net
    proto = tmp_path / "pose_deploy_linevec.prototxt"
    pose_net.write_prototxt(proto)
    assert pose_net.names_from_prototxt(proto) == pose_net.caffe_layer_names()
    convs, pools = pose_net.layers_from_prototxt(proto)
    assert len(convs) == 92 and pools == ["conv1_2", "conv2_2", "conv3_4"]
    assert [c[1:5] for c in convs] == [l[1:] for l in pose_net.all_layers()]     # cin (185 behind the Concat layers), cout, k, relu
    assert pose_net.check_prototxt(proto) == 92
    # renamed layers, written in reverse order with ReLU entries in between: found by the names of the description
    renamed = ["L%02d_%s" % (i, n) for i, n in enumerate(pose_net.caffe_layer_names())]
    proto2 = tmp_path / "renamed.prototxt"
    pose_net.write_prototxt(proto2, names=renamed)
    assert pose_net.names_from_prototxt(proto2) == renamed
    layers = [(cn, ci, co, k) for (_, ci, co, k, _), cn in zip(pose_net.all_layers(), renamed)]
    model = tmp_path / "renamed.caffemodel"
    _write_model(model, layers[::-1])
    assert pose_net.check_prototxt(proto2, model) == 92
    with pytest.raises(ValueError, match="L00_conv1_1|no weights"):
        pose_net.check_prototxt(proto, model)            # the published names are not in that file
    # the published file's layout: the two branches of a stage interleaved layer by layer (conv5_1_CPM_L1, conv5_1_CPM_L2,
    # conv5_2_CPM_L1 ...) -- the layers are found by walking the blobs, not by their position in the file
    inter, inter2 = tmp_path / "interleaved.prototxt", tmp_path / "interleaved_renamed.prototxt"
    pose_net.write_prototxt(inter, interleaved=True)
    lines = inter.read_text().splitlines()
    at = next(i for i, ln in enumerate(lines) if '"conv5_1_CPM_L1"' in ln and "Convolution" in ln)
    assert '"conv5_1_CPM_L2"' in lines[at + 2] and '"conv5_2_CPM_L1"' in lines[at + 4]
    assert pose_net.names_from_prototxt(inter) == pose_net.caffe_layer_names() and pose_net.check_prototxt(inter) == 92
    pose_net.write_prototxt(inter2, names=renamed, interleaved=True)
    assert pose_net.names_from_prototxt(inter2) == renamed and pose_net.check_prototxt(inter2, model) == 92
    # ... with the L2 branch written first, and a ReLU that is not in place
    swapped = tmp_path / "swapped.prototxt"
    blocks = inter.read_text().splitlines()
    l1 = next(i for i, ln in enumerate(blocks) if '"Mconv1_stage4_L1"' in ln and "Convolution" in ln)
    blocks[l1:l1 + 4] = blocks[l1 + 2:l1 + 4] + blocks[l1:l1 + 2]
    txt = "\n".join(blocks).replace('name: "relu_conv3_2" type: "ReLU" bottom: "conv3_2" top: "conv3_2"', 'name: "relu_conv3_2" type: "ReLU" bottom: "conv3_2" top: "conv3_2_relu"')
    txt = txt.replace('name: "conv3_3" type: "Convolution" bottom: "conv3_2"', 'name: "conv3_3" type: "Convolution" bottom: "conv3_2_relu"')
    assert "conv3_2_relu" in txt
    swapped.write_text(txt + "\n")
    assert pose_net.names_from_prototxt(swapped) == pose_net.caffe_layer_names() and pose_net.check_prototxt(swapped) == 92
    # the stage input concatenated in another order than the weights are packed for: refused, not silently mis-bound
    catswap = tmp_path / "catswap.prototxt"
    catswap.write_text(inter.read_text().replace('bottom: "conv5_5_CPM_L1" bottom: "conv5_5_CPM_L2" bottom: "conv4_4_CPM"',
                                                 'bottom: "conv5_5_CPM_L2" bottom: "conv5_5_CPM_L1" bottom: "conv4_4_CPM"'))
    assert catswap.read_text() != inter.read_text()
    for check in (pose_net.names_from_prototxt, pose_net.check_prototxt):
        with pytest.raises(ValueError, match="order"):
            check(catswap)
    # another network: one layer fewer / another kernel size / another width
    text = proto.read_text()
    short = tmp_path / "short.prototxt"
    short.write_text("\n".join(ln for ln in text.splitlines() if '"Mconv7_stage6_L2"' not in ln))
    for check in (pose_net.names_from_prototxt, pose_net.check_prototxt):
        with pytest.raises(ValueError, match="91"):
            check(short)
    k5 = tmp_path / "k5.prototxt"
    k5.write_text(text.replace('name: "Mconv2_stage3_L1" type: "Convolution" bottom: "Mconv1_stage3_L1" top: "Mconv2_stage3_L1" convolution_param { num_output: 128 pad: 3 kernel_size: 7 }',
                               'name: "Mconv2_stage3_L1" type: "Convolution" bottom: "Mconv1_stage3_L1" top: "Mconv2_stage3_L1" convolution_param { num_output: 128 pad: 2 kernel_size: 5 }'))
    assert k5.read_text() != text
    for check in (pose_net.names_from_prototxt, pose_net.check_prototxt):
        with pytest.raises(ValueError, match="Mconv2_stage3_L1"):
            check(k5)
    wide = tmp_path / "wide.prototxt"
    wide.write_text(text.replace('top: "conv4_4_CPM" convolution_param { num_output: 128', 'top: "conv4_4_CPM" convolution_param { num_output: 96'))
    with pytest.raises(ValueError, match="conv4_4_CPM"):
        pose_net.check_prototxt(wide)
    with pytest.raises(ValueError):
        pose_net.names_from_prototxt(wide)
    # text-format details: comments, the colon before a message, single quotes, V1 upper-case types
    fancy = tmp_path / "fancy.prototxt"
    fancy.write_text("# a comment\n" + text.replace("convolution_param {", "convolution_param: {", 3).replace('"Convolution"', "'Convolution'", 2))
    assert pose_net.names_from_prototxt(fancy) == pose_net.caffe_layer_names() and pose_net.check_prototxt(fancy) == 92
    # malformed descriptions are ValueErrors, never a hang or a KeyError: two renaming ReLUs in a cycle, a layer without blobs
    cyc = tmp_path / "cycle.prototxt"
    cyc.write_text(text + '\nlayer { name: "ra" type: "ReLU" bottom: "xa" top: "xb" }\nlayer { name: "rb" type: "ReLU" bottom: "xb" top: "xa" }\n'
                          'layer { name: "cc" type: "Concat" bottom: "xa" bottom: "xb" top: "xc" }\n')
    with pytest.raises(ValueError, match="cycle"):
        pose_net.names_from_prototxt(cyc)
    for frag in ('layer { name: "rn" type: "ReLU" bottom: "conv1_1" }', 'layer { name: "pn" type: "Pooling" top: "zz" }',
                 'layer { name: "cn" type: "Concat" bottom: "conv1_1" }'):
        nob = tmp_path / "noblob.prototxt"
        nob.write_text(text + "\n" + frag + "\n")
        for check in (pose_net.names_from_prototxt, pose_net.check_prototxt):
            with pytest.raises(ValueError, match="blob"):
                check(nob)
    broken = tmp_path / "broken.prototxt"
    broken.write_text(text[:len(text) // 2])
    with pytest.raises(ValueError):
        pose_net.check_prototxt(broken)
    with pytest.raises(ValueError, match="cannot read"):
        pose_net.check_prototxt(tmp_path / "absent.prototxt")
