"""GPU: the convolution stack of the pose network (MFMA kernels, csrc/st_conv.hip) against
torch.nn.functional on the same float32 weights -- layer shapes one by one (every kernel size, both block
widths, channel slices, pixel-count tails) and the whole network end to end."""
import ctypes
import os

import numpy as np
import pytest
import torch

from scannertools_amd import pose_net

pytestmark = pytest.mark.gpu


# the float32 matrix instruction (default), the opt-in split-bf16 arithmetic of the same accuracy (3x3 / 7x7 layers with
# 128-channel output blocks on the spatial-tile kernel, the rest on the per-tap kernel), and the latter for every layer
MATH = ["f32", "bf16x3", "f32_pertap", "bf16x3_pertap", "f32_tile4", "bf16x3_tile4"]   # _tile4: the 4-wave tile instances
NET_MATH = MATH[:2]   # PoseNet(math=...)
_PERTAP = {}


def _mode_ctx(value):
    """A context created under ST_CONV_TILE=value (read at st_ctx_create): "0" the per-tap kernels for every layer, "1" the
    spatial-tile kernels wherever a tile shape exists (left to itself the library picks by launch size -- the same bits)."""
    if value not in _PERTAP:
        from scannertools_amd.hip import HipContext
        saved = os.environ.get("ST_CONV_TILE")
        os.environ["ST_CONV_TILE"] = value
        try:
            _PERTAP[value] = HipContext(0)
        finally:
            if saved is None:
                del os.environ["ST_CONV_TILE"]
            else:
                os.environ["ST_CONV_TILE"] = saved
    return _PERTAP[value]


def _pertap_ctx():
    return _mode_ctx("0")


def _conv(hip_ctx, x_nhwc, cin, xoff, wt, b, relu, cout_total=None, yoff=0, math="f32"):
    """x_nhwc (n,h,w,C) cuda; wt (co,ci,k,k), b (co,) cpu -> y (n,h,w,cout_total) cuda."""
    n, h, w, xs = x_nhwc.shape
    co, ci, k, _ = wt.shape
    cip = (ci + 15) // 16 * 16
    assert cip == cin
    cop = (co + 63) // 64 * 64
    wp = torch.zeros((cop, k, k, cip))
    wp[:co, :, :, :ci] = wt.permute(0, 2, 3, 1)
    bp = torch.zeros((cop,))
    bp[:co] = b
    wp, bp = wp.cuda(), bp.cuda()
    ys = cout_total or (co + 3) // 4 * 4
    y = torch.full((n, h, w, ys), -7.0, dtype=torch.float32, device="cuda")
    if math.endswith("_pertap"):
        hip_ctx, math = _pertap_ctx(), math[:-7]
    elif math.endswith("_tile41"):
        hip_ctx, math = _mode_ctx("41"), math[:-7]   # 4 waves x one 32-pixel instruction tile (bf16x3; float32: the 4-wave instance)
    elif math.endswith("_tile4"):
        hip_ctx, math = _mode_ctx("4"), math[:-6]
    else:
        hip_ctx = _mode_ctx("1")   # the tile kernels whatever the launch size
    hip_ctx._bind()
    if math == "bf16x3":
        w3 = torch.empty((hip_ctx._L.st_conv_bf16x3_packed_bytes(cop, k, k, cip),), dtype=torch.uint8, device="cuda")
        hip_ctx._check(hip_ctx._L.st_conv_pack_weights_bf16x3(hip_ctx._h, ctypes.c_void_p(wp.data_ptr()), cop, k, k, cip, ctypes.c_void_p(w3.data_ptr())))
        hip_ctx._check(hip_ctx._L.st_conv2d_nhwc_bf16x3(hip_ctx._h, ctypes.c_void_p(x_nhwc.data_ptr()), n, h, w, cin, xs, xoff,
                                                         ctypes.c_void_p(w3.data_ptr()), ctypes.c_void_p(bp.data_ptr()), k, k, co, cop,
                                                         int(relu), ctypes.c_void_p(y.data_ptr()), ys, yoff))
        return y
    nb = hip_ctx._L.st_conv_f32_tile_bytes(cop, k, k, cip)
    wt_ = None
    if nb > 0:
        wt_ = torch.empty((nb,), dtype=torch.uint8, device="cuda")
        hip_ctx._check(hip_ctx._L.st_conv_pack_weights_f32_tile(hip_ctx._h, ctypes.c_void_p(wp.data_ptr()), cop, k, k, cip, ctypes.c_void_p(wt_.data_ptr())))
    hip_ctx._check(hip_ctx._L.st_conv2d_nhwc_f32_tiled(hip_ctx._h, ctypes.c_void_p(x_nhwc.data_ptr()), n, h, w, cin, xs, xoff,
                                                        ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(wt_.data_ptr()) if wt_ is not None else None,
                                                        ctypes.c_void_p(bp.data_ptr()), k, k, co, cop,
                                                        int(relu), ctypes.c_void_p(y.data_ptr()), ys, yoff))
    return y


@pytest.mark.parametrize("n,h,w,ci,co,k,relu", [(1, 9, 13, 3, 64, 3, 1), (2, 23, 31, 16, 128, 3, 1), (1, 17, 19, 128, 38, 1, 0),
                                                 (2, 12, 20, 64, 19, 7, 0), (1, 46, 82, 185, 128, 7, 1), (3, 8, 8, 512, 512, 3, 1),
                                                 (1, 5, 7, 32, 200, 5, 1),
                                                 # the spatial-tile kernel's edges: one-pixel maps, a map narrower than the kernel, tiles
                                                 # cut by the right / bottom border, two tile columns, two output blocks, one slice
                                                 (2, 1, 1, 16, 128, 7, 1), (1, 3, 2, 32, 128, 3, 0), (2, 47, 83, 48, 128, 7, 1),
                                                 (1, 50, 300, 32, 256, 3, 1), (1, 33, 130, 16, 100, 7, 0)])
@pytest.mark.parametrize("math", MATH)
def test_conv_layer_matches_torch(hip_ctx, n, h, w, ci, co, k, relu, math):
    g = torch.Generator().manual_seed(n * 1000 + ci + co + k)
    x = torch.randn((n, ci, h, w), generator=g)
    wt = torch.randn((co, ci, k, k), generator=g) * float(np.sqrt(2.0 / (ci * k * k)))
    b = torch.randn((co,), generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), padding=k // 2)
    if relu:
        ref = torch.relu(ref)
    cip = (ci + 15) // 16 * 16
    xn = torch.zeros((n, h, w, cip))
    xn[..., :ci] = x.permute(0, 2, 3, 1)
    y = _conv(hip_ctx, xn.cuda(), cip, 0, wt, b, relu, math=math)
    got = y[..., :co].permute(0, 3, 1, 2).cpu().double()
    scale = float(ref.abs().max())
    # the same bound for both: against float64, the split-bf16 products must be as good as float32 products
    assert float((got - ref).abs().max()) <= 2e-5 * max(scale, 1.0), (float((got - ref).abs().max()), scale)
    rel = float((got - ref).norm() / ref.norm())
    assert rel <= 2e-6, rel
    assert (y[..., co:] == -7.0).all()                       # nothing written outside the cout channels


@pytest.mark.parametrize("math", MATH)
def test_conv_channel_slices_and_identity_known_answer(hip_ctx, math):
    """Reads a channel slice of a wider buffer, writes into a slice of another; a 1x1 identity kernel and a
    3x3 shift kernel have known answers (asymmetric, so a transposed operand layout cannot pass)."""
    n, h, w = 1, 11, 14
    g = torch.Generator().manual_seed(5)
    buf = torch.randn((n, h, w, 48), generator=g)
    x = buf[..., 16:32]                                       # the slice the layer reads (offset 16)
    wt = torch.zeros((16, 16, 1, 1))
    for c in range(16):
        wt[c, (c + 3) % 16, 0, 0] = 1.0                       # out channel c = in channel c+3: a permutation, not symmetric
    y = _conv(hip_ctx, buf.cuda(), 16, 16, wt, torch.zeros(16), 0, cout_total=40, yoff=20, math=math)
    got = y.cpu()
    np.testing.assert_array_equal(got[..., 20:36].numpy(), x[..., [(c + 3) % 16 for c in range(16)]].numpy())
    assert (got[..., :20] == -7).all() and (got[..., 36:] == -7).all()
    wt3 = torch.zeros((16, 16, 3, 3))
    for c in range(16):
        wt3[c, c, 0, 2] = 1.0                                 # tap (ky=0, kx=2): out(y, x) = in(y-1, x+1), zero outside
    y = _conv(hip_ctx, buf.cuda(), 16, 16, wt3, torch.zeros(16), 0, math=math)
    exp = torch.zeros((n, h, w, 16))
    exp[:, 1:, :-1] = x[:, :-1, 1:]
    np.testing.assert_array_equal(y[..., :16].cpu().numpy(), exp.numpy())


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
def test_conv_tile_random_shapes(hip_ctx, math):
    """Seeded random map sizes, channel counts and batch sizes for the spatial-tile kernels (3x3 / 7x7, 128-channel output blocks):
    every tile shape the planner picks -- tiles cut by the right and bottom borders, several tile columns, maps smaller than a
    tile, inputs read from a channel slice -- against float64, and bit for bit against the per-tap kernel of the same
    arithmetic (both accumulate slices outer, taps inner, so the launcher may pick either by launch size)."""
    rng = np.random.default_rng(20260 + len(math))
    for case in range(24):
        k = (3, 7)[case % 2]
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 140))
        ci = int(rng.choice([16, 32, 48, 80]))
        co = int(rng.choice([128, 100, 256, 129]))
        n = int(rng.integers(1, 4))
        xoff = int(rng.choice([0, 16]))
        g = torch.Generator().manual_seed(case)
        x = torch.randn((n, h, w, ci + xoff + 16), generator=g)
        wt = torch.randn((co, ci, k, k), generator=g) * float(np.sqrt(2.0 / (ci * k * k)))
        b = torch.randn((co,), generator=g) * 0.1
        ref = torch.relu(torch.nn.functional.conv2d(x[..., xoff:xoff + ci].permute(0, 3, 1, 2).double(), wt.double(), b.double(), padding=k // 2))
        ys = (co + 3) // 4 * 4 + 8
        y = _conv(hip_ctx, x.cuda(), ci, xoff, wt, b, 1, cout_total=ys, yoff=4, math=math)
        yp = _conv(hip_ctx, x.cuda(), ci, xoff, wt, b, 1, cout_total=ys, yoff=4, math=math + "_pertap")
        got = y[..., 4:4 + co].permute(0, 3, 1, 2).cpu().double()
        scale = max(float(ref.abs().max()), 1.0)
        msg = (case, n, h, w, ci, co, k)
        assert float((got - ref).abs().max()) <= 2e-5 * scale, msg
        assert torch.equal(y, yp), msg    # the per-tap kernels accumulate in the tile kernels' order: the same bits
        assert torch.equal(y, _conv(hip_ctx, x.cuda(), ci, xoff, wt, b, 1, cout_total=ys, yoff=4, math=math + "_tile4")), msg
        assert torch.equal(y, _conv(hip_ctx, x.cuda(), ci, xoff, wt, b, 1, cout_total=ys, yoff=4, math=math + "_tile41")), msg
        assert (y[..., :4] == -7.0).all() and (y[..., 4 + co:] == -7.0).all(), msg


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
@pytest.mark.parametrize("mode", ["1", "4", "0", "auto"])
def test_conv_pair_equals_two_calls(hip_ctx, math, mode):
    """st_conv2d_nhwc_*_pair -- two convolutions of one geometry, different operands, in one launch where the tile kernel runs
    (the two branches of a stage) -- gives exactly what the two single calls give: different inputs / channel slices,
    different weights, different output counts under one cout_pad, every kernel choice."""
    from scannertools_amd._native import ConvOperands
    ctx = hip_ctx if mode == "auto" else _mode_ctx(mode)
    for (n, h, w, ci, k, co_a, co_b) in ((2, 46, 82, 32, 7, 128, 128), (3, 23, 31, 48, 3, 100, 128), (1, 9, 13, 16, 1, 38, 19), (2, 12, 20, 16, 7, 38, 19)):
        g = torch.Generator().manual_seed(h * w + k)
        cop = (max(co_a, co_b) + 63) // 64 * 64
        xa = torch.randn((n, h, w, ci + 16), generator=g).cuda()
        xb = torch.randn((n, h, w, ci), generator=g).cuda()
        ys, ws, ops, keep = [], [], [], []
        for x, xoff, co, yoff in ((xa, 16, co_a, 0), (xb, 0, co_b, 4)):
            wt = torch.zeros((cop, k, k, ci))
            wt[:co] = torch.randn((co, k, k, ci), generator=g) * float(np.sqrt(2.0 / (ci * k * k)))
            b = torch.zeros((cop,))
            b[:co] = torch.randn((co,), generator=g) * 0.1
            wt, b = wt.cuda(), b.cuda()
            ctx._bind()
            if math == "bf16x3":
                wq = torch.empty((ctx._L.st_conv_bf16x3_packed_bytes(cop, k, k, ci),), dtype=torch.uint8, device="cuda")
                ctx._check(ctx._L.st_conv_pack_weights_bf16x3(ctx._h, ctypes.c_void_p(wt.data_ptr()), cop, k, k, ci, ctypes.c_void_p(wq.data_ptr())))
                wmain, wtile = wq, None
            else:
                nb = ctx._L.st_conv_f32_tile_bytes(cop, k, k, ci)
                wtile = torch.empty((nb,), dtype=torch.uint8, device="cuda") if nb else None
                if nb:
                    ctx._check(ctx._L.st_conv_pack_weights_f32_tile(ctx._h, ctypes.c_void_p(wt.data_ptr()), cop, k, k, ci, ctypes.c_void_p(wtile.data_ptr())))
                wmain = wt
            y1 = torch.full((n, h, w, co + 8), -7.0, device="cuda")
            y2 = torch.full((n, h, w, co + 8), -7.0, device="cuda")
            vp = ctypes.c_void_p
            if math == "bf16x3":
                ctx._check(ctx._L.st_conv2d_nhwc_bf16x3(ctx._h, vp(x.data_ptr()), n, h, w, ci, x.shape[3], xoff, vp(wmain.data_ptr()), vp(b.data_ptr()),
                                                        k, k, co, cop, 1, vp(y1.data_ptr()), co + 8, yoff))
            else:
                ctx._check(ctx._L.st_conv2d_nhwc_f32_tiled(ctx._h, vp(x.data_ptr()), n, h, w, ci, x.shape[3], xoff, vp(wmain.data_ptr()),
                                                           vp(wtile.data_ptr()) if wtile is not None else None, vp(b.data_ptr()),
                                                           k, k, co, cop, 1, vp(y1.data_ptr()), co + 8, yoff))
            o = ConvOperands()
            o.x, o.x_stride, o.x_offset = x.data_ptr(), x.shape[3], xoff
            o.w, o.w_tile = wmain.data_ptr(), (wtile.data_ptr() if wtile is not None else None)
            o.bias, o.cout = b.data_ptr(), co
            o.y, o.y_stride, o.y_offset = y2.data_ptr(), co + 8, yoff
            ys.append((y1, y2)); ops.append(o); keep.append((wt, b, wmain, wtile))
        fn = ctx._L.st_conv2d_nhwc_bf16x3_pair if math == "bf16x3" else ctx._L.st_conv2d_nhwc_f32_pair
        ctx._check(fn(ctx._h, n, h, w, ci, k, k, cop, 1, ctypes.byref(ops[0]), ctypes.byref(ops[1])))
        torch.cuda.synchronize()
        for y1, y2 in ys:
            assert torch.equal(y1, y2), (n, h, w, ci, k, co_a, co_b)
    # refusals: a null operand set, a bad slice in the second one
    assert fn(ctx._h, 1, 4, 4, 16, 3, 3, 64, 1, ctypes.byref(ops[0]), None) != 0
    bad = ConvOperands.from_buffer_copy(ops[1])
    bad.y_offset = 10 ** 6
    assert fn(ctx._h, n, h, w, ci, k, k, cop, 1, ctypes.byref(ops[0]), ctypes.byref(bad)) != 0


def test_maxpool_and_layout(hip_ctx):
    g = torch.Generator().manual_seed(2)
    x = torch.randn((2, 3, 10, 14), generator=g).cuda()
    nhwc = torch.empty((2, 10, 14, 16), device="cuda")
    hip_ctx._bind()
    hip_ctx._check(hip_ctx._L.st_planar_to_nhwc_f32(hip_ctx._h, ctypes.c_void_p(x.data_ptr()), 2, 3, 10, 14,
                                                     ctypes.c_void_p(nhwc.data_ptr()), 16))
    assert torch.equal(nhwc[..., :3], x.permute(0, 2, 3, 1)) and (nhwc[..., 3:] == 0).all()
    y = torch.empty((2, 5, 7, 16), device="cuda")
    hip_ctx._check(hip_ctx._L.st_maxpool2_nhwc_f32(hip_ctx._h, ctypes.c_void_p(nhwc.data_ptr()), 2, 10, 14, 16, 16,
                                                    ctypes.c_void_p(y.data_ptr()), 16))
    ref = torch.nn.functional.max_pool2d(x, 2).permute(0, 2, 3, 1)
    assert torch.equal(y[..., :3], ref)


@pytest.mark.parametrize("math", NET_MATH)
def test_pose_network_end_to_end(hip_ctx, math):
    """All 92 convolutions + 3 poolings on a small input, against the float32 torch network (CPU, so that no
    other GPU library is in the comparison); the reference's own precision, so the bound is float32 round-off
    accumulated over the depth."""
    net = pose_net.PoseNet(hip_ctx, seed=3, math=math)
    g = torch.Generator().manual_seed(9)
    x = (torch.rand((2, 3, 48, 80), generator=g) - 0.5)
    got = net.forward(x.cuda()).permute(0, 3, 1, 2).cpu()
    ref = net.reference_forward(x, device="cpu")
    assert got.shape == ref.shape == (2, 57, 6, 10)
    scale = float(ref.abs().max())
    assert scale > 1e-3
    assert float((got - ref).abs().max()) <= 1e-3 * scale, (float((got - ref).abs().max()), scale)
    assert pose_net.flops(368, 656) > 4e11 and len(pose_net.all_layers()) == 92


@pytest.mark.parametrize("math", NET_MATH)
def test_network_bits_do_not_depend_on_the_kernel_choice(hip_ctx, math):
    """The library picks the per-tap or the spatial-tile kernel per layer by launch size; both accumulate in the same order,
    so a frame's maps are the same bits whichever runs -- alone, in a batch of 6, with the tile kernels forced, with the
    per-tap kernels forced."""
    g = torch.Generator().manual_seed(33)
    x = torch.rand((6, 3, 96, 128), generator=g) - 0.5
    outs = {}
    for name, ctx in (("auto", hip_ctx), ("tile", _mode_ctx("1")), ("pertap", _mode_ctx("0")), ("tile4", _mode_ctx("4")), ("tile41", _mode_ctx("41"))):
        net = pose_net.PoseNet(ctx, seed=8, math=math)
        outs[name] = net.forward(x.cuda()).cpu()
        if name == "auto":
            outs["single"] = net.forward(x[2:3].cuda()).cpu()
        del net
    assert torch.equal(outs["auto"], outs["tile"]) and torch.equal(outs["auto"], outs["pertap"]) and torch.equal(outs["auto"], outs["tile4"]) and torch.equal(outs["auto"], outs["tile41"])
    assert torch.equal(outs["single"][0], outs["auto"][2])


def test_whole_pose_pipeline_through_the_engine(hip_ctx):
    """frames -> CPM2Input -> CPM2 (network + resize + nms, random weights) -> CPM2Output -> poses, as one graph
    (the op chain of the reference's pose pipeline: cpm2_input_kernel_gpu.cpp:184, cpm2_kernel.cpp:46-52,
    cpm2_output_kernel_cpu.cpp:805-810).  Checked link by link: the op's two columns equal the oracle's `resize`
    and `nms` restatements applied to the network's own output, and the poses equal the oracle's assembly of those
    columns."""
    import oracle
    from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, NamedVideoStream, PerfParams
    from util import random_frames
    frames = random_frames(11, 3, 135, 240)
    scale = 64 / 135.
    sc = Client()
    sc.ingest_frames("v", frames)
    frame = sc.io.Input([NamedVideoStream(sc, "v")])
    net_in = sc.ops.CPM2Input(frame=frame, scale=scale, device=DeviceType.GPU, batch=2)
    maps_col, joints_col = sc.ops.CPM2(cpm2_input=net_in, seed=4, batch=2)
    poses = sc.ops.CPM2Output(cpm2_resized_map=maps_col, cpm2_joints=joints_col, original_frame_info=sc.ops.InfoFromFrame(frame=frame),
                              scale=scale, device=DeviceType.GPU, batch=2)
    out = NamedStream(sc, "poses")
    sc.run(sc.io.Output(poses, [out]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
    got = list(out.load())
    assert len(got) == 3

    net = pose_net.PoseNet(hip_ctx, seed=4)
    x = torch.from_numpy(np.stack([oracle.cpm2_input(f, scale) for f in frames])).cuda()
    low = net.forward(x).cpu().numpy()                       # (n, h/8, w/8, 57)
    maps, joints = net.detect(x)
    H, W = x.shape[2], x.shape[3]
    for i in range(3):
        ref_maps = oracle.cpm2_resize_maps(np.ascontiguousarray(low[i].transpose(2, 0, 1)), H, W)
        np.testing.assert_array_equal(maps[i].cpu().numpy(), ref_maps)
        ref_joints = oracle.cpm2_nms(ref_maps, 18, 64, 0.05)
        np.testing.assert_array_equal(joints[i].cpu().numpy(), ref_joints)
        np.testing.assert_array_equal(got[i], oracle.cpm2_connect_limbs_coco(ref_maps, ref_joints, 135, 240))
    assert float(joints[:, :, 0, 0].max()) > 0   # the random network does produce candidates
    # the kernel classes record their work under the keys the reference's kernels use with Scanner's Profiler
    # (cpm2_input_kernel_gpu.cpp:153-155): one interval per execute() call, 3 rows at batch 2 = 2 calls
    assert sc.profile["cpm2_input"][0] == 2 and 0 < sc.profile["cpm2_input"][1] < 5.0


@pytest.fixture(scope="module")
def model_dir(hip_ctx, tmp_path_factory):
    """An OpenPose model directory (<dir>/pose/coco/pose_iter_440000.caffemodel, openpose_kernel.cpp:47-52) holding the
    weights of PoseNet(seed=6), written with the wire-format helpers: (directory, caffemodel path, the network)."""
    a = pose_net.PoseNet(hip_ctx, seed=6)
    root = tmp_path_factory.mktemp("openpose_models")
    path = root / "pose" / "coco" / "pose_iter_440000.caffemodel"
    path.parent.mkdir(parents=True)
    pose_net.write_caffemodel(str(path), a.weights)
    return str(root), str(path), a


def test_pose_net_loads_a_caffemodel(hip_ctx, model_dir, tmp_path):
    """PoseNet(caffemodel=...) == PoseNet with the same weights set directly (a complete 92-layer file; tiny spatial
    check, the weights are what is being tested), and so do the registered kernel classes."""
    _, path, a = model_dir
    assert pose_net.check_caffemodel(path) == 92
    b_net = pose_net.PoseNet(hip_ctx, caffemodel=path)
    x = (torch.rand((3, 3, 16, 24), generator=torch.Generator().manual_seed(1)) - 0.5).cuda()
    assert torch.equal(a.forward(x), b_net.forward(x))

    # the registered kernel classes (CPM2KernelHIP on DeviceType::GPU, its staged twin on DeviceType::CPU) read the
    # same file and produce the same two columns as PoseNet.detect, bit for bit
    from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, PerfParams
    maps, joints = a.detect(x)

    class _Rows:
        def __init__(self, rows):
            self.rows_ = rows

        def length(self):
            return len(self.rows_)

        def rows(self, idx):
            return [self.rows_[i] for i in idx]

    for device in (DeviceType.GPU, DeviceType.CPU):
        sc = Client()
        src = _Rows([f for f in (x if device == DeviceType.GPU else x.cpu().numpy())])
        m_col, j_col = sc.ops.CPM2(cpm2_input=src, weights=path, device=device, batch=2)
        om, oj = NamedStream(sc, "maps"), NamedStream(sc, "joints")
        sc.run([sc.io.Output(m_col, [om]), sc.io.Output(j_col, [oj])], PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
        for i, (m, j) in enumerate(zip(om.load(), oj.load())):
            np.testing.assert_array_equal(m, maps[i].cpu().numpy())
            np.testing.assert_array_equal(j, joints[i].cpu().numpy())
        # "caffe:net" intervals (caffe_kernel.cpp:381-387): one per execute() of each of the two columns' runs
        assert sc.profile["caffe:net"][0] >= 2 and sc.profile["caffe:net"][1] > 0

    # the opt-in split-bf16 arithmetic: the kernel class (SCANNERTOOLS_POSE_MATH, read when the instance is created)
    # equals PoseNet(math="bf16x3") bit for bit, and the two arithmetics agree to float32 round-off over the 92 layers
    import os
    c_net = pose_net.PoseNet(hip_ctx, caffemodel=path, math="bf16x3")
    low32, low3 = a.forward(x), c_net.forward(x)
    assert float((low32 - low3).abs().max()) <= 2e-5 * float(low32.abs().max()) and not torch.equal(low32, low3)
    maps3, joints3 = c_net.detect(x)
    os.environ["SCANNERTOOLS_POSE_MATH"] = "bf16x3"
    try:
        sc = Client()
        m_col, j_col = sc.ops.CPM2(cpm2_input=_Rows([f for f in x]), weights=path, device=DeviceType.GPU, batch=2)
        om, oj = NamedStream(sc, "maps3"), NamedStream(sc, "joints3")
        sc.run([sc.io.Output(m_col, [om]), sc.io.Output(j_col, [oj])], PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
        for i, (m, j) in enumerate(zip(om.load(), oj.load())):
            np.testing.assert_array_equal(m, maps3[i].cpu().numpy())
            np.testing.assert_array_equal(j, joints3[i].cpu().numpy())
    finally:
        del os.environ["SCANNERTOOLS_POSE_MATH"]
    # a value that names no arithmetic of the build is a validation error of the kernel instance, not a silent float32
    os.environ["SCANNERTOOLS_POSE_MATH"] = "fp8"
    try:
        with pytest.raises(RuntimeError, match="SCANNERTOOLS_POSE_MATH"):
            sc = Client()
            m_col, _ = sc.ops.CPM2(cpm2_input=_Rows([x[0]]), weights=path, device=DeviceType.GPU)
            sc.run(sc.io.Output(m_col, [NamedStream(sc, "m_bad")]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
    finally:
        del os.environ["SCANNERTOOLS_POSE_MATH"]

    cut = tmp_path / "cut.caffemodel"
    with open(path, "rb") as fh:
        cut.write_bytes(fh.read(100 * 1000 * 1000))
    with pytest.raises(ValueError):
        pose_net.PoseNet(hip_ctx, caffemodel=str(cut))
    with pytest.raises(RuntimeError, match="CPM2"):
        sc = Client()
        m_col, _ = sc.ops.CPM2(cpm2_input=_Rows([x[0]]), weights=str(cut), device=DeviceType.GPU)
        sc.run(sc.io.Output(m_col, [NamedStream(sc, "m")]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)


def test_model_with_other_layer_order_and_names(hip_ctx, model_dir, tmp_path):
    """A caffemodel whose layers are stored in ANOTHER order than the compiled-in list (reversed, ReLU entries in between)
    under OTHER names, described by its own prototxt: PoseNet and the CPM2 kernel class take the names from the
    description, find every layer, and compute the same bits as the network built from the weights directly."""
    from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, PerfParams
    _, _, a = model_dir
    renamed = ["net_%s_%02d" % (n, i) for i, n in enumerate(pose_net.caffe_layer_names())]
    model, proto = tmp_path / "shuffled.caffemodel", tmp_path / "shuffled.prototxt"
    order = list(range(92))[::-1]
    pose_net.write_caffemodel(str(model), a.weights, names=renamed, order=order, extra_layers=["relu_%d" % i for i in range(40)])
    pose_net.write_prototxt(str(proto), names=renamed)
    assert pose_net.check_prototxt(proto, model) == 92
    with pytest.raises(ValueError):
        pose_net.PoseNet(hip_ctx, caffemodel=str(model))              # the published names are not in this file
    b_net = pose_net.PoseNet(hip_ctx, caffemodel=str(model), prototxt=str(proto))
    x = (torch.rand((2, 3, 16, 24), generator=torch.Generator().manual_seed(4)) - 0.5).cuda()
    assert torch.equal(a.forward(x), b_net.forward(x))
    maps, joints = a.detect(x)

    class _Rows:
        def __init__(self, rows):
            self.rows_ = rows

        def length(self):
            return len(self.rows_)

        def rows(self, idx):
            return [self.rows_[i] for i in idx]

    sc = Client()
    m_col, j_col = sc.ops.CPM2(cpm2_input=_Rows([f for f in x]), weights=str(model), prototxt=str(proto), device=DeviceType.GPU, batch=2)
    om, oj = NamedStream(sc, "maps_s"), NamedStream(sc, "joints_s")
    sc.run([sc.io.Output(m_col, [om]), sc.io.Output(j_col, [oj])], PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
    for i, (m, j) in enumerate(zip(om.load(), oj.load())):
        np.testing.assert_array_equal(m, maps[i].cpu().numpy())
        np.testing.assert_array_equal(j, joints[i].cpu().numpy())
    with pytest.raises(RuntimeError, match="CPM2"):                    # without the description the kernel class looks for the published names
        sc = Client()
        m_col, _ = sc.ops.CPM2(cpm2_input=_Rows([x[0]]), weights=str(model), device=DeviceType.GPU)
        sc.run(sc.io.Output(m_col, [NamedStream(sc, "m_no_proto")]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)


@pytest.mark.parametrize("S", [1, 2, 3])
def test_resize_merge_maps_bit_exact(hip_ctx, S):
    """The scales' maps merged on the GPU == the oracle's sum of per-scale interpolants / S; one scale == the single-scale
    kernel bit for bit (both with identity and with the pose channel map)."""
    import oracle
    rng = np.random.default_rng(40 + S)
    dims = [(46, 82), (40, 70), (33, 58)][:S]
    maps = [rng.standard_normal((2, h, w, 192)).astype(np.float32) for h, w in dims]
    eff = [(46.0, 82.0), (46 * 0.85, 82 * 0.853), (46 * 0.7, 82 * 0.71)][:S]
    chan = [pose_net.OFF_HEAT + i for i in range(19)] + [pose_net.OFF_PAF + i for i in range(38)]
    cu = [torch.from_numpy(m).cuda() for m in maps]
    got = hip_ctx.cpm2_resize_merge_maps(cu, eff, 368, 656, chan_map=chan).cpu().numpy()
    for i in range(2):
        ref = oracle.cpm2_resize_merge_maps([np.ascontiguousarray(m[i].transpose(2, 0, 1)[chan]) for m in maps], eff, 368, 656)
        np.testing.assert_array_equal(got[i], ref)
    if S == 1:
        single = hip_ctx.cpm2_resize_maps(cu[0], 368, 656, chan_map=chan)
        assert torch.equal(single, torch.from_numpy(got).cuda())
        assert np.array_equal(np.signbit(single.cpu().numpy()), np.signbit(got))


@pytest.mark.parametrize("scales,gap,device", [(1, 0.0, "gpu"), (3, 0.15, "gpu"), (2, 0.25, "cpu")])
def test_openpose_op(hip_ctx, model_dir, scales, gap, device):
    """sc.ops.OpenPose through the registered kernel classes == the same chain assembled from the Python-visible pieces
    (CPM2Input per scale -> network -> merge -> nms -> limb scores -> the oracle's assembly -> the element layout of
    openpose_kernel.cpp:175-212), byte for byte; hands / faces and a missing model are refused."""
    import oracle
    from scannertools_amd import pose_detection
    from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedStream, NamedVideoStream, PerfParams
    from scannertools_amd.hip import cpm2_geometry, cpm2_scale_for_height
    from util import random_frames
    root, _, net = model_dir
    frames = random_frames(17, 3, 96, 160)
    H, W = 96, 160
    sc = Client()
    sc.ingest_frames("v", frames)
    frame = sc.io.Input([NamedVideoStream(sc, "v")])
    dev = DeviceType.GPU if device == "gpu" else DeviceType.CPU
    out = NamedStream(sc, "pose")
    sc.run(sc.io.Output(sc.ops.OpenPose(frame=frame, model_directory=root, pose_num_scales=scales, pose_scale_gap=gap, device=dev, batch=2),
                        [out]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
    raw = sc._tables["pose"][0]
    people = list(out.load())

    fr = torch.from_numpy(frames).cuda()
    geo, raws = [], []
    for i in range(scales):
        target = int(np.floor(np.float32(368) * (np.float32(1) - np.float32(i) * np.float32(gap)) + np.float32(0.5)))
        s = cpm2_scale_for_height(H, target)
        geo.append(cpm2_geometry(H, W, s))
        assert geo[-1][0] == target
        raws.append(net.forward_raw(hip_ctx.cpm2_input(fr, s)))
    (rh0, rw0, nh0, nw0) = geo[0]
    f32 = np.float32
    eff = [(float(nh0 // 8), float(nw0 // 8))] + [(float(f32(nh0 // 8) * (f32(g[0]) / f32(rh0))), float(f32(nw0 // 8) * (f32(g[1]) / f32(rw0)))) for g in geo[1:]]
    chan = [pose_net.OFF_HEAT + i for i in range(19)] + [pose_net.OFF_PAF + i for i in range(38)]
    maps = hip_ctx.cpm2_resize_merge_maps(raws, eff, nh0, nw0, chan_map=chan)
    joints = hip_ctx.cpm2_nms(maps, parts=18, max_peaks=64, threshold=0.05)
    scores = hip_ctx.cpm2_limb_scores(maps, joints).cpu().numpy()
    total = 0
    for i in range(3):
        # joints of the oracle's assembly in network-input pixels (frame = network input), then ZeroToOne
        pj = oracle.cpm2_connect_limbs_coco(maps[i].cpu().numpy(), joints[i].cpu().numpy(), nh0, nw0, scores=scores[i])
        if len(pj) == 0:
            assert raw[i] == b"\0\0\0\0" and people[i] == []
            continue
        total += len(pj)
        rec = np.zeros((len(pj), pose_detection.Pose.kp_size()), np.float32)
        kp = pj.copy()
        kp[:, :, 0] = kp[:, :, 0] / f32(rw0)
        kp[:, :, 1] = kp[:, :, 1] / f32(rh0)
        acc = np.zeros(len(pj), np.float32)
        for j in range(18):
            acc = acc + kp[:, j, 2]
        rec[:, 0] = acc / f32(18)
        rec[:, 1:1 + 54] = kp.reshape(len(pj), 54)
        assert raw[i] == rec.tobytes()
        assert len(people[i]) == len(pj) and people[i][0].pose_keypoints().shape == (18, 3)
        assert float(np.abs(people[i][0].face_keypoints()).max()) == 0.0
    print("people assembled from the random network:", total)

    if scales == 1 and device == "gpu":
        for kw, msg in ((dict(compute_hands=True), "hand"), (dict(compute_face=True), "face"), (dict(model_directory=""), "model_directory"),
                        (dict(model_directory=root + "/nowhere"), "cannot read"), (dict(pose_num_scales=4, pose_scale_gap=0.4), "pose_num_scales")):
            args = dict(frame=frame, model_directory=root, device=dev)
            args.update(kw)
            with pytest.raises(RuntimeError, match=msg):
                sc.run(sc.io.Output(sc.ops.OpenPose(**args), [NamedStream(sc, "bad")]), PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
