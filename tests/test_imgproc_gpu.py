"""GPU parity of the sibling imgproc ops: Blur (HIP, through the C ABI and the
Scanner kernel class) vs the CPU oracle -- bit-exact (integer arithmetic spelled out in the
reference source, blur_kernel_cpu.cpp:62-79)."""
import numpy as np
import pytest
import torch

import oracle
from scannertools_amd._native import COLOR_CODES as COLOR_CODES_ALL
from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedVideoStream, PerfParams
from util import random_frames, texture_stream

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 7, 8, 9, 11, 15, 31])
@pytest.mark.parametrize("h,w", [(37, 53), (64, 344), (120, 683)])
def test_box_blur_matches_oracle(hip_ctx, h, w, k):
    frames = random_frames(h + w + k, 3, h, w)
    got = hip_ctx.box_blur(torch.from_numpy(frames).cuda(), k).cpu().numpy()
    assert got.dtype == np.uint8 and got.shape == frames.shape
    for i in range(3):
        np.testing.assert_array_equal(got[i], oracle.box_blur(frames[i], k))


@pytest.mark.parametrize("h,w", [(1, 1), (2, 3), (5, 4), (480, 640), (1080, 1920)])
def test_box_blur_shapes(hip_ctx, h, w):
    """Frames smaller than the window (no interior at all), the reference test clip's size, 1080p;
    rows that are not dword aligned (3*w % 4 != 0)."""
    frames = random_frames(h * w, 2, h, w)
    for k in (3, 7, 9, 11):
        got = hip_ctx.box_blur(torch.from_numpy(frames).cuda(), k).cpu().numpy()
        for i in range(2):
            np.testing.assert_array_equal(got[i], oracle.box_blur(frames[i], k))


def test_box_blur_known_answers_and_errors(hip_ctx):
    from scannertools_amd._native import StError
    h, w = 40, 50
    const = np.full((1, h, w, 3), 200, np.uint8)
    got = hip_ctx.box_blur(torch.from_numpy(const).cuda(), 5).cpu().numpy()[0]
    assert (got[2:-2, 2:-2] == 200).all() and got[:2].sum() == 0 and got[:, :2].sum() == 0     # border = 0
    assert got[-2:].sum() == 0 and got[:, -2:].sum() == 0
    # even kernel sizes are asymmetric: left = k/2 - 1, right = k/2 (blur_kernel_cpu.cpp:38-39)
    got = hip_ctx.box_blur(torch.from_numpy(const).cuda(), 4).cpu().numpy()[0]
    assert (got[1:-2, 1:-2] == 200).all() and got[0].sum() == 0 and got[-2:].sum() == 0
    one = torch.from_numpy(const).cuda()
    with pytest.raises(StError):
        hip_ctx.box_blur(one, 0)
    with pytest.raises(StError):
        hip_ctx.box_blur(one, 33)
    with pytest.raises(StError):
        hip_ctx.box_blur(one, 3, out=one)                      # in place is refused
    with pytest.raises(TypeError):
        hip_ctx.box_blur(torch.from_numpy(const), 3)           # CPU tensor: no fallback


@pytest.mark.parametrize("device", [DeviceType.CPU, DeviceType.GPU])
def test_blur_op_like_the_reference_test(device):
    """scannertools/tests/test_all.py:180-194 replayed (plus the value check it lacks)."""
    sc = Client()
    frames, _ = texture_stream(3, 40, 96, 128)
    sc.ingest_frames('test1', frames)
    input = NamedVideoStream(sc, 'test1')
    frame = sc.io.Input([input])
    range_frame = sc.streams.Range(frame, ranges=[{'start': 0, 'end': 30}])
    blurred_frame = sc.ops.Blur(frame=range_frame, kernel_size=3, sigma=0.1, device=device, batch=7)
    output = NamedVideoStream(sc, 'test_blur')
    output_op = sc.io.Output(blurred_frame, [output])
    sc.run(output_op, PerfParams.estimate(), cache_mode=CacheMode.Overwrite, show_progress=False)

    frame_array = next(output.load())
    assert frame_array.dtype == np.uint8
    assert frame_array.shape[0] == 96
    assert frame_array.shape[1] == 128
    assert frame_array.shape[2] == 3
    loaded = list(output.load())
    assert len(loaded) == 30
    for i, f in enumerate(loaded):
        np.testing.assert_array_equal(f, oracle.box_blur(frames[i], 3))
    assert sc.live_device_buffers() == 0


# ---- Resize ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("sh,sw,dh,dw", [(480, 640, 240, 426), (1080, 1920, 240, 426), (48, 64, 24, 32),
                                         (37, 53, 80, 91), (100, 100, 1, 1), (9, 7, 9, 7), (64, 48, 16, 12),
                                         (33, 47, 66, 94), (240, 426, 1080, 1920)])
def test_resize_linear_matches_oracle(hip_ctx, sh, sw, dh, dw):
    """Down- and up-scaling, the legacy pipeline's 426x240, exact 2x (INTER_AREA mean), exact 4x
    (stays linear), identity."""
    frames = random_frames(sh + sw + dh, 2, sh, sw)
    got = hip_ctx.resize(torch.from_numpy(frames).cuda(), dw, dh).cpu().numpy()
    assert got.shape == (2, dh, dw, 3) and got.dtype == np.uint8
    for i in range(2):
        np.testing.assert_array_equal(got[i], oracle.resize_u8(frames[i], dw, dh))


@pytest.mark.parametrize("sh,sw,dh,dw", [(480, 640, 240, 426), (48, 60, 16, 20), (48, 60, 24, 30), (48, 60, 13, 21),
                                         (37, 53, 80, 91), (48, 60, 20, 90), (48, 60, 100, 13), (100, 100, 1, 1),
                                         (1080, 1920, 360, 640), (9, 7, 9, 7), (270, 480, 180, 320), (50, 70, 33, 47), (50, 70, 26, 36),
                                         (50, 70, 49, 69), (50, 71, 30, 36), (48, 60, 12, 15), (48, 60, 24, 15), (48, 60, 8, 20), (90, 120, 30, 24)])
@pytest.mark.parametrize("interp", ["cubic", "area", "lanczos4"])
def test_resize_cubic_and_area_match_oracle(hip_ctx, sh, sw, dh, dw, interp):
    """INTER_CUBIC; INTER_LANCZOS4; INTER_AREA with integer cells (3x, 2x), fractional cells,
    enlargement on one or both axes, identity."""
    from scannertools_amd._native import INTER_AREA, INTER_CUBIC, INTER_LANCZOS4
    code, ocode = {"cubic": (INTER_CUBIC, oracle.INTER_CUBIC), "area": (INTER_AREA, oracle.INTER_AREA),
                   "lanczos4": (INTER_LANCZOS4, oracle.INTER_LANCZOS4)}[interp]
    frames = random_frames(sh + sw + dw, 2, sh, sw)
    got = hip_ctx.resize(torch.from_numpy(frames).cuda(), dw, dh, code).cpu().numpy()
    for i in range(2):
        np.testing.assert_array_equal(got[i], oracle.resize_u8(frames[i], dw, dh, ocode))


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_resize_channels_nearest_and_identities(hip_ctx, cn):
    from scannertools_amd._native import INTER_NEAREST, StError
    rng = np.random.default_rng(cn)
    f = rng.integers(0, 256, (2, 90, 120, cn), dtype=np.uint8)
    for (dw, dh) in ((50, 40), (240, 180), (60, 45)):
        got = hip_ctx.resize(torch.from_numpy(f).cuda(), dw, dh).cpu().numpy()
        np.testing.assert_array_equal(got[1], oracle.resize_u8(f[1], dw, dh))
        got = hip_ctx.resize(torch.from_numpy(f).cuda(), dw, dh, INTER_NEAREST).cpu().numpy()
        np.testing.assert_array_equal(got[0], oracle.resize_u8(f[0], dw, dh, oracle.INTER_NEAREST))
    const = np.full((1, 33, 71, cn), 201, np.uint8)
    assert (hip_ctx.resize(torch.from_numpy(const).cuda(), 19, 100).cpu().numpy() == 201).all()
    with pytest.raises(StError):
        hip_ctx.resize(torch.from_numpy(f).cuda(), 10, 10, interpolation=7)      # INTER_MAX: a mask, not a mode
    from scannertools_amd._native import INTER_LANCZOS4
    got = hip_ctx.resize(torch.from_numpy(f).cuda(), 77, 31, INTER_LANCZOS4).cpu().numpy()
    np.testing.assert_array_equal(got[1], oracle.resize_u8(f[1], 77, 31, oracle.INTER_LANCZOS4))


@pytest.mark.parametrize("device", [DeviceType.CPU, DeviceType.GPU])
def test_legacy_flow_histogram_pipeline(device):
    """old/histograms.py:63-78: Resize(426x240) -> OpticalFlow -> FlowHistogram, every stage checked."""
    sc = Client()
    frames, _ = texture_stream(11, 6, 480, 640, max_step=4)
    sc.ingest_frames('v', frames)
    frame = sc.io.Input([NamedVideoStream(sc, 'v')])
    small = sc.ops.Resize(frame=frame, width=426, height=240, device=device, batch=4)
    flow = sc.ops.OpticalFlow(frame=small, device=device, batch=4)
    fh = sc.ops.FlowHistogram(flow=flow, device=device)
    from scannertools_amd.engine import NamedStream
    o_small, o_flow, o_fh = NamedStream(sc, 's'), NamedStream(sc, 'f'), NamedStream(sc, 'h')
    sc.run([sc.io.Output(small, [o_small]), sc.io.Output(flow, [o_flow]), sc.io.Output(fh, [o_fh])],
           PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
    smalls = list(o_small.load())
    assert len(smalls) == 6 and smalls[0].shape == (240, 426, 3)
    for i, s in enumerate(smalls):
        np.testing.assert_array_equal(s, oracle.resize_u8(frames[i], 426, 240))
    for i, (fl, h) in enumerate(zip(o_flow.load(), o_fh.load())):
        ref = oracle.optical_flow_rgb(smalls[i], smalls[min(i + 1, 5)])
        assert np.abs(fl - ref).max() <= 5e-3
        np.testing.assert_array_equal(np.stack(h), oracle.flow_hist(fl))


def test_resize_op_target_size_rules():
    """preserve_aspect / min of ResizeArgs (resize_kernel.cpp:44-62) through the kernel class."""
    sc = Client()
    frames = random_frames(1, 3, 120, 200)
    sc.ingest_frames('v', frames)
    frame = sc.io.Input([NamedVideoStream(sc, 'v')])
    from scannertools_amd.engine import NamedStream
    cases = [dict(width=100, height=0, preserve_aspect=True), dict(width=0, height=30, preserve_aspect=True),
             dict(width=400, height=300, min=True), dict(width=50, height=60, min=True),
             dict(width=64, height=32, interpolation="INTER_NEAREST")]
    for i, kw in enumerate(cases):
        out = NamedStream(sc, 'r%d' % i)
        sc.run(sc.io.Output(sc.ops.Resize(frame=frame, device=DeviceType.GPU, **kw), [out]), PerfParams.estimate(),
               cache_mode=CacheMode.Overwrite)
        tw, th = oracle.resize_target(200, 120, kw.get("width", 0), kw.get("height", 0), kw.get("min", False),
                                      kw.get("preserve_aspect", False))
        interp = oracle.INTER_NEAREST if kw.get("interpolation") == "INTER_NEAREST" else oracle.INTER_LINEAR
        for j, got in enumerate(out.load()):
            assert got.shape == (th, tw, 3)
            np.testing.assert_array_equal(got, oracle.resize_u8(frames[j], tw, th, interp))


# ---- ConvertColor -----------------------------------------------------------------------------------
@pytest.mark.parametrize("name,code", [("COLOR_BGR2RGB", 4), ("COLOR_RGB2BGR", 4), ("COLOR_BGR2GRAY", 6),
                                       ("COLOR_RGB2GRAY", 7), ("COLOR_BGR2HSV", 40), ("COLOR_BGR2YCrCb", 36),
                                       ("COLOR_RGB2YCrCb", 37), ("COLOR_YCrCb2BGR", 38), ("COLOR_YCrCb2RGB", 39),
                                       ("COLOR_RGB2HSV", 41), ("COLOR_HSV2BGR", 54), ("COLOR_HSV2RGB", 55),
                                       ("COLOR_BGR2HSV_FULL", 66), ("COLOR_RGB2HSV_FULL", 67), ("COLOR_HSV2BGR_FULL", 70),
                                       ("COLOR_HSV2RGB_FULL", 71), ("COLOR_BGR2YUV", 82), ("COLOR_RGB2YUV", 83),
                                       ("COLOR_YUV2BGR", 84), ("COLOR_YUV2RGB", 85), ("COLOR_BGR2XYZ", 32), ("COLOR_RGB2XYZ", 33),
                                       ("COLOR_XYZ2BGR", 34), ("COLOR_XYZ2RGB", 35),
                                       ("COLOR_BGR2HLS", 52), ("COLOR_RGB2HLS", 53), ("COLOR_HLS2BGR", 60), ("COLOR_HLS2RGB", 61),
                                       ("COLOR_BGR2HLS_FULL", 68), ("COLOR_RGB2HLS_FULL", 69), ("COLOR_HLS2BGR_FULL", 72),
                                       ("COLOR_HLS2RGB_FULL", 73)])
@pytest.mark.parametrize("h,w", [(1, 1), (6, 10), (37, 53), (480, 640)])
def test_cvt_color_matches_oracle(hip_ctx, name, code, h, w):
    frames = random_frames(h + w + code, 2, h, w)
    got = hip_ctx.cvt_color(torch.from_numpy(frames).cuda(), name).cpu().numpy()
    for i in range(2):
        np.testing.assert_array_equal(got[i], oracle.cvt_color(frames[i], code))
    if "GRAY" in name:
        g14 = hip_ctx.cvt_color(torch.from_numpy(frames).cuda(), code, gray_bits=14).cpu().numpy()
        np.testing.assert_array_equal(g14[0], oracle.cvt_color(frames[0], code, gray_bits=14))


def test_cvt_color_exhaustive_hsv_and_known_answers(hip_ctx):
    """Every (b,g,r) with 5-bit components + all saturated ramps through BGR2HSV; primaries."""
    v = (np.arange(32) * 8 + 3).astype(np.uint8)
    grid = np.stack(np.meshgrid(v, v, v, indexing="ij"), -1).reshape(1, 32, 1024, 3)
    got = hip_ctx.cvt_color(torch.from_numpy(np.ascontiguousarray(grid)).cuda(), "COLOR_BGR2HSV").cpu().numpy()
    np.testing.assert_array_equal(got[0], oracle.cvt_color(grid[0], oracle.COLOR_BGR2HSV))
    px = np.array([[[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [128, 128, 128]]]], np.uint8)
    hsv = hip_ctx.cvt_color(torch.from_numpy(px).cuda(), "COLOR_BGR2HSV").cpu().numpy()[0, 0]
    assert hsv.tolist() == [[120, 255, 255], [60, 255, 255], [0, 255, 255], [0, 0, 255], [0, 0, 0], [0, 0, 128]]
    gray = hip_ctx.cvt_color(torch.from_numpy(px).cuda(), "COLOR_BGR2GRAY").cpu().numpy()[0, 0, :, 0]
    assert gray.tolist() == [29, 150, 76, 255, 0, 128]
    g = np.random.default_rng(0).integers(0, 256, (1, 9, 11, 1), dtype=np.uint8)
    rgb = hip_ctx.cvt_color(torch.from_numpy(g).cuda(), "COLOR_GRAY2RGB").cpu().numpy()
    assert rgb.shape == (1, 9, 11, 3) and (rgb == g).all()
    # the whole 8-bit cube (stride 3 per component + the top value) through the codes whose
    # arithmetic is not a plain integer table: HSV -> RGB runs in float on both sides
    v = np.unique(np.concatenate([np.arange(0, 256, 3), [254, 255]])).astype(np.uint8)
    cube = np.ascontiguousarray(np.stack(np.meshgrid(v, v, v, indexing="ij"), -1).reshape(1, len(v), len(v) * len(v), 3))
    for name in ("COLOR_HSV2BGR", "COLOR_HSV2RGB_FULL", "COLOR_RGB2HSV", "COLOR_BGR2HSV_FULL", "COLOR_RGB2YUV", "COLOR_YUV2BGR",
                 "COLOR_BGR2XYZ", "COLOR_RGB2XYZ", "COLOR_XYZ2BGR", "COLOR_XYZ2RGB",
                 # HLS runs in float both ways (divisions, a data-dependent branch per pixel): the cube must agree bit for bit
                 "COLOR_BGR2HLS", "COLOR_RGB2HLS_FULL", "COLOR_HLS2BGR", "COLOR_HLS2RGB", "COLOR_HLS2BGR_FULL", "COLOR_HLS2RGB_FULL"):
        got = hip_ctx.cvt_color(torch.from_numpy(cube).cuda(), name).cpu().numpy()
        np.testing.assert_array_equal(got[0], oracle.cvt_color(cube[0], getattr(oracle, name)), err_msg=name)
    hls = hip_ctx.cvt_color(torch.from_numpy(px).cuda(), "COLOR_BGR2HLS").cpu().numpy()[0, 0]
    assert hls.tolist() == [[120, 128, 255], [60, 128, 255], [0, 128, 255], [0, 255, 0], [0, 0, 0], [0, 128, 0]]
    xyz = hip_ctx.cvt_color(torch.from_numpy(px).cuda(), "COLOR_BGR2XYZ").cpu().numpy()[0, 0]
    assert xyz[3].tolist() == [242, 255, 255] and xyz[4].tolist() == [0, 0, 0]      # white (Z saturates), black
    with pytest.raises(ValueError):
        hip_ctx.cvt_color(torch.from_numpy(px).cuda(), "COLOR_BGR2Lab")
    with pytest.raises(ValueError):
        hip_ctx.cvt_color(torch.from_numpy(g).cuda(), "COLOR_BGR2HSV")             # needs 3 channels


def test_cvt_color_layout_family_exhaustive(hip_ctx):
    """cv::cvtColor codes 0..3, 5, 9..31 (alpha channel added / dropped / swapped, BGR565 / BGR555 packed pixels, gray from /
    to them): every code on every value its input can take (all 65 536 packed pixels; the byte cube for 3- and 4-channel
    sources), bit for bit against the oracle's per-functor restatement, plus the identities the packing implies."""
    from scannertools_amd._native import COLOR_CODES
    v = np.unique(np.concatenate([np.arange(0, 256, 5), [1, 2, 3, 4, 6, 7, 8, 248, 251, 252, 253, 254, 255]])).astype(np.uint8)
    cube3 = np.ascontiguousarray(np.stack(np.meshgrid(v, v, v, indexing="ij"), -1).reshape(1, len(v), -1, 3))
    alpha = (np.arange(cube3.shape[1] * cube3.shape[2]).reshape(1, cube3.shape[1], -1, 1) * 37 % 256).astype(np.uint8)
    alpha[0, 0, :7, 0] = 0                      # some fully transparent pixels (the 555 alpha bit)
    cube4 = np.ascontiguousarray(np.concatenate([cube3, alpha], axis=3))
    packed = np.arange(65536, dtype="<u2").view(np.uint8).reshape(1, 256, 256, 2)
    gray = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1)
    sources = {1: gray, 2: np.ascontiguousarray(packed), 3: cube3, 4: cube4}
    names = [n for n, c in COLOR_CODES.items() if c <= 3 or c == 5 or 9 <= c <= 31]
    assert len(names) == 34 and len({COLOR_CODES[n] for n in names}) == 28
    for name in names:
        code = COLOR_CODES[name]
        cin = next(c for c in (1, 2, 3, 4) if oracle.lib().orc_cvt_out_channels(code, c) > 0)
        src = sources[cin]
        got = hip_ctx.cvt_color(torch.from_numpy(src).cuda(), name).cpu().numpy()
        np.testing.assert_array_equal(got[0], oracle.cvt_color(src[0], code), err_msg=name)
        for wrong in (1, 2, 3, 4):
            if wrong != cin:
                with pytest.raises(ValueError):
                    hip_ctx.cvt_color(torch.from_numpy(sources[wrong]).cuda(), name)
    cu = lambda a: torch.from_numpy(a).cuda()
    # identities: alpha added then dropped; swap twice; packing keeps the top 5 / 6 / 5 bits; 565 -> gray == gray of the unpacked pixel (14-bit table)
    a4 = hip_ctx.cvt_color(cu(cube3), "COLOR_BGR2BGRA")
    assert int(a4[..., 3].min()) == 255 and torch.equal(hip_ctx.cvt_color(a4, "COLOR_BGRA2BGR"), cu(cube3))
    assert torch.equal(hip_ctx.cvt_color(hip_ctx.cvt_color(cu(cube4), "COLOR_BGRA2RGBA"), "COLOR_RGBA2BGRA"), cu(cube4))
    p565 = hip_ctx.cvt_color(cu(cube3), "COLOR_BGR2BGR565")
    assert torch.equal(hip_ctx.cvt_color(p565, "COLOR_BGR5652BGR"), cu(cube3 & np.array([0xF8, 0xFC, 0xF8], np.uint8)))
    assert torch.equal(hip_ctx.cvt_color(p565, "COLOR_BGR5652RGB"), cu(np.ascontiguousarray((cube3 & np.array([0xF8, 0xFC, 0xF8], np.uint8))[..., ::-1])))
    p555 = hip_ctx.cvt_color(cu(cube4), "COLOR_BGRA2BGR555")
    back = hip_ctx.cvt_color(p555, "COLOR_BGR5552BGRA").cpu().numpy()
    np.testing.assert_array_equal(back[..., :3], cube4[..., :3] & 0xF8)
    np.testing.assert_array_equal(back[..., 3], np.where(cube4[..., 3] != 0, 255, 0))
    g = hip_ctx.cvt_color(p565, "COLOR_BGR5652GRAY")
    assert torch.equal(g, hip_ctx.cvt_color(hip_ctx.cvt_color(p565, "COLOR_BGR5652BGR"), "COLOR_BGR2GRAY", gray_bits=14))


@pytest.mark.parametrize("device", [DeviceType.CPU, DeviceType.GPU])
def test_hsv_histogram_pipeline(device):
    """old/histograms.py:21-40 (compute_hsv_histograms): colour conversion -> Histogram, plus a
    gray conversion whose single-channel output changes the frame shape."""
    from scannertools_amd.engine import NamedStream
    sc = Client()
    frames = random_frames(9, 5, 60, 80)
    sc.ingest_frames('v', frames)
    frame = sc.io.Input([NamedVideoStream(sc, 'v')])
    hsv = sc.ops.ConvertColor(frame=frame, conversion='COLOR_RGB2HSV', device=device, batch=3)   # old/cpp_ops/imgproc.cpp:41
    hist = sc.ops.Histogram(frame=hsv, device=device, batch=2)
    gray = sc.ops.ConvertColor(frame=frame, conversion='COLOR_RGB2GRAY', device=device)
    o_hist, o_gray = NamedStream(sc, 'hh'), NamedStream(sc, 'gg')
    sc.run([sc.io.Output(hist, [o_hist]), sc.io.Output(gray, [o_gray])], PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
    for i, (hh, gg) in enumerate(zip(o_hist.load(), o_gray.load())):
        np.testing.assert_array_equal(np.stack(hh), oracle.hist_u8c3(oracle.cvt_color(frames[i], oracle.COLOR_RGB2HSV), 16))
        assert gg.shape == (60, 80, 1)
        np.testing.assert_array_equal(gg, oracle.cvt_color(frames[i], oracle.COLOR_RGB2GRAY))


@pytest.mark.parametrize("H,W", [(2, 2), (6, 10), (10, 20), (8, 48), (54, 98), (270, 480)])
def test_cvt_color_yuv_sources(hip_ctx, H, W):
    """cv::cvtColor codes 90..124: NV12 / NV21 / YV12 / IYUV 4:2:0 frames ((3H/2, W, 1), heights that are and are not
    multiples of 4) and UYVY / YUY2 / YVYU 4:2:2 frames ((H, W, 2)) to RGB / BGR / RGBA / BGRA / gray, every name, against the
    oracle's block-wise restatement; BT.601 known answers; shapes that cannot be such frames are refused."""
    from scannertools_amd._native import COLOR_CODES
    from util import cvt_source
    rng = np.random.default_rng(H * W)
    names = [n for n, c in COLOR_CODES.items() if 90 <= c <= 124]
    assert len(names) == 71 and len({COLOR_CODES[n] for n in names}) == 31
    for name in names:
        code = COLOR_CODES[name]
        src = np.stack([cvt_source(rng, code, H, W) for _ in range(2)])
        if rng.random() < 0.5:                      # extreme chroma / luma: the saturating branches
            src[0] = rng.choice(np.array([0, 15, 16, 17, 128, 234, 235, 236, 255], np.uint8), src[0].shape)
        got = hip_ctx.cvt_color(torch.from_numpy(src).cuda(), name).cpu().numpy()
        for i in range(2):
            np.testing.assert_array_equal(got[i], oracle.cvt_color(src[i], code), err_msg=name)
        assert got.shape[1:3] == (H, W)
    # BT.601: (Y, U, V) = (235, 128, 128) white, (16, 128, 128) black, (81, 90, 240) red, (145, 54, 34) green, (41, 240, 110) blue
    for (yy, uu, vv), rgb in (((235, 128, 128), (255, 255, 255)), ((16, 128, 128), (0, 0, 0)), ((81, 90, 240), (254, 0, 0)),
                               ((145, 54, 34), (0, 255, 1)), ((41, 240, 110), (0, 0, 255))):
        nv12 = np.zeros((1, H * 3 // 2, W, 1), np.uint8)
        nv12[0, :H] = yy
        nv12[0, H:, 0::2] = uu
        nv12[0, H:, 1::2] = vv
        out = hip_ctx.cvt_color(torch.from_numpy(nv12).cuda(), "COLOR_YUV2RGB_NV12").cpu().numpy()
        assert np.abs(out.reshape(-1, 3).astype(int) - np.array(rgb)).max() <= 1, (yy, uu, vv, out[0, 0, 0])
        assert (out == out[0, 0, 0]).all()
        i420 = np.concatenate([np.full(H * W, yy), np.full(H * W // 4, uu), np.full(H * W // 4, vv)]).astype(np.uint8).reshape(1, H * 3 // 2, W, 1)
        assert np.array_equal(hip_ctx.cvt_color(torch.from_numpy(i420).cuda(), "COLOR_YUV2RGB_I420").cpu().numpy(), out)
    with pytest.raises(ValueError):
        hip_ctx.cvt_color(torch.zeros((1, 4, 6, 3), dtype=torch.uint8, device="cuda"), "COLOR_YUV2RGB_NV12")     # not single-channel
    with pytest.raises(ValueError):
        hip_ctx.cvt_color(torch.zeros((1, 4, 6, 1), dtype=torch.uint8, device="cuda"), "COLOR_YUV2RGB_NV12")     # height not 3H/2
    with pytest.raises(ValueError):
        hip_ctx.cvt_color(torch.zeros((1, 6, 5, 1), dtype=torch.uint8, device="cuda"), "COLOR_YUV2RGB_NV12")     # odd width
    with pytest.raises(ValueError):
        hip_ctx.cvt_color(torch.zeros((1, 4, 5, 2), dtype=torch.uint8, device="cuda"), "COLOR_YUV2BGR_YUY2")     # odd width


def test_cvt_color_unaligned_frames(hip_ctx):
    """Frames that start on odd addresses (a view into a larger buffer) take the byte-wise kernels; 4-byte aligned ones with a
    multiple of 4 pixels the 4-pixel kernels; results are the same."""
    rng = np.random.default_rng(11)
    for name, shape in (("COLOR_RGB2HSV", (16, 32, 3)), ("COLOR_RGB2BGR565", (16, 32, 3)), ("COLOR_YUV2RGB_NV12", (24, 32, 1)),
                        ("COLOR_YUV2BGRA_UYVY", (16, 32, 2)), ("COLOR_BGRA2GRAY", (16, 32, 4))):
        src = rng.integers(0, 256, shape, dtype=np.uint8)
        ref = oracle.cvt_color(src, COLOR_CODES_ALL[name])
        nbytes = src.size
        for off in (0, 1, 4, 6):
            buf = torch.zeros(nbytes + 32, dtype=torch.uint8, device="cuda")
            view = buf[off:off + nbytes].view(1, *shape)
            view.copy_(torch.from_numpy(src))
            obuf = torch.zeros(ref.size + 32, dtype=torch.uint8, device="cuda")
            out = obuf[off:off + ref.size].view(1, *ref.shape)
            hip_ctx.cvt_color(view, name, out=out)
            np.testing.assert_array_equal(out[0].cpu().numpy(), ref, err_msg="%s offset %d" % (name, off))
            assert int(obuf[:off].sum()) == 0 and int(obuf[off + ref.size:].sum()) == 0     # nothing written outside the frame


def test_resize_linear_fast_path_alignments(hip_ctx):
    """INTER_LINEAR on 3-channel frames takes the 4-columns-per-thread kernel (unaligned dword loads and stores): widths that
    are and are not multiples of 4, source and destination frames on odd addresses, single-column and single-row
    targets, enlargement -- bit for bit the oracle's values and nothing written outside the frame."""
    from scannertools_amd._native import INTER_LINEAR
    rng = np.random.default_rng(21)
    for (h, w, dh, dw) in ((37, 53, 20, 31), (37, 53, 21, 32), (64, 96, 100, 150), (9, 7, 1, 1), (9, 7, 3, 1), (9, 7, 1, 5), (120, 200, 67, 113)):
        src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = oracle.resize_u8(src, dw, dh, INTER_LINEAR)
        for off in (0, 1, 3):
            sbuf = torch.zeros(src.size + 16, dtype=torch.uint8, device="cuda")
            sview = sbuf[off:off + src.size].view(1, h, w, 3)
            sview.copy_(torch.from_numpy(src))
            obuf = torch.zeros(ref.size + 16, dtype=torch.uint8, device="cuda")
            out = obuf[off:off + ref.size].view(1, dh, dw, 3)
            hip_ctx.resize(sview, dw, dh, INTER_LINEAR, out=out)
            np.testing.assert_array_equal(out[0].cpu().numpy(), ref, err_msg="%dx%d -> %dx%d offset %d" % (h, w, dh, dw, off))
            assert int(obuf[:off].sum()) == 0 and int(obuf[off + ref.size:].sum()) == 0


def test_resize_exact_half_fast_path(hip_ctx):
    """The exact 2x2 decimation (INTER_AREA and INTER_LINEAR's reroute) on 3-channel frames: 4 output pixels per thread with a
    partial last group, odd addresses; other channel counts keep the generic kernel; all bit for bit the oracle's."""
    from scannertools_amd._native import INTER_AREA, INTER_LINEAR
    rng = np.random.default_rng(23)
    for (h, w, cn) in ((20, 30, 3), (64, 96, 3), (2, 2, 3), (2, 10, 3), (18, 26, 1), (18, 26, 4)):
        src = rng.integers(0, 256, (h, w, cn), dtype=np.uint8)
        for interp in (INTER_LINEAR, INTER_AREA):
            ref = oracle.resize_u8(src, w // 2, h // 2, interp)
            for off in (0, 1):
                sbuf = torch.zeros(src.size + 16, dtype=torch.uint8, device="cuda")
                sview = sbuf[off:off + src.size].view(1, h, w, cn)
                sview.copy_(torch.from_numpy(src))
                obuf = torch.zeros(ref.size + 16, dtype=torch.uint8, device="cuda")
                out = obuf[off:off + ref.size].view(1, h // 2, w // 2, cn)
                hip_ctx.resize(sview, w // 2, h // 2, interp, out=out)
                np.testing.assert_array_equal(out[0].cpu().numpy(), ref, err_msg="%dx%dx%d interp %d offset %d" % (h, w, cn, interp, off))
                assert int(obuf[:off].sum()) == 0 and int(obuf[off + ref.size:].sum()) == 0


def test_nv12_ingest_pipeline():
    """Decoder-style ingest: NV12 frames -> ConvertColor(COLOR_YUV2RGB_NV12) -> Histogram, through the
    kernel classes; the output-shape probe turns (3H/2, W, 1) frames into (H, W, 3) ones."""
    from scannertools_amd.engine import NamedStream
    rng = np.random.default_rng(8)
    H, W = 48, 64
    nv12 = rng.integers(0, 256, (4, H * 3 // 2, W, 1), dtype=np.uint8)
    sc = Client()
    sc.ingest_frames('nv12', nv12)
    frame = sc.io.Input([NamedVideoStream(sc, 'nv12')])
    for device in (DeviceType.GPU, DeviceType.CPU):
        rgb = sc.ops.ConvertColor(frame=frame, conversion='COLOR_YUV2RGB_NV12', device=device, batch=3)
        hist = sc.ops.Histogram(frame=rgb, device=device, batch=2)
        o_rgb, o_hist = NamedStream(sc, 'rgb'), NamedStream(sc, 'hist')
        sc.run([sc.io.Output(rgb, [o_rgb]), sc.io.Output(hist, [o_hist])], PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
        for i, (r, hh) in enumerate(zip(o_rgb.load(), o_hist.load())):
            ref = oracle.cvt_color(nv12[i], 90)
            assert r.shape == (H, W, 3)
            np.testing.assert_array_equal(r, ref)
            np.testing.assert_array_equal(np.stack(hh), oracle.hist_u8c3(ref, 16))


@pytest.mark.parametrize("device", [DeviceType.CPU, DeviceType.GPU])
def test_layout_conversions_through_the_kernel_classes(device):
    """ConvertColor with output frames of 4 and 2 channels, chained (RGB -> RGBA -> BGR565 -> gray): the op's
    output-shape probe (convert_color_kernel.cpp:252-277) follows the conversion."""
    from scannertools_amd.engine import NamedStream
    sc = Client()
    frames = random_frames(12, 4, 33, 47)
    sc.ingest_frames('v', frames)
    frame = sc.io.Input([NamedVideoStream(sc, 'v')])
    rgba = sc.ops.ConvertColor(frame=frame, conversion='COLOR_RGB2RGBA', device=device, batch=3)
    packed = sc.ops.ConvertColor(frame=rgba, conversion='COLOR_RGBA2BGR565', device=device, batch=2)
    gray = sc.ops.ConvertColor(frame=packed, conversion='COLOR_BGR5652GRAY', device=device)
    o = [NamedStream(sc, n) for n in ('a', 'p', 'g')]
    sc.run([sc.io.Output(rgba, [o[0]]), sc.io.Output(packed, [o[1]]), sc.io.Output(gray, [o[2]])], PerfParams.estimate(), cache_mode=CacheMode.Overwrite)
    for i, (a, p, g) in enumerate(zip(o[0].load(), o[1].load(), o[2].load())):
        ra = oracle.cvt_color(frames[i], 0)
        rp = oracle.cvt_color(ra, 17)
        assert a.shape == (33, 47, 4) and p.shape == (33, 47, 2) and g.shape == (33, 47, 1)
        np.testing.assert_array_equal(a, ra)
        np.testing.assert_array_equal(p, rp)
        np.testing.assert_array_equal(g, oracle.cvt_color(rp, 21))


def test_empty_batches_and_bad_handles(hip_ctx):
    """n = 0 is a no-op for every entry point; a null context is an error code, not a crash."""
    import ctypes
    from scannertools_amd import _native
    assert hip_ctx.box_blur([], 3).shape[0] == 0
    assert hip_ctx.resize([], 4, 4).shape[0] == 0
    assert hip_ctx.cvt_color([], "COLOR_BGR2GRAY").shape[0] == 0
    assert hip_ctx.draw_flow([], []).shape[0] == 0
    assert hip_ctx.flow_histogram([]).shape == (0, 2, 64)
    L = _native.lib()
    null = ctypes.c_void_p()
    tab = (ctypes.c_void_p * 1)()
    assert L.st_box_blur_u8c3_batch(null, tab, 1, 4, 4, 3, tab) != 0
    assert L.st_resize_u8_batch(null, tab, 1, 4, 4, 3, 2, 2, 1, tab) != 0
    assert L.st_cvt_color_u8_batch(null, tab, 1, 4, 4, 3, 6, 15, tab) != 0
    assert L.st_flow_hist_batch(null, tab, 1, 4, 4, None) != 0
    assert L.st_draw_flow_batch(null, tab, tab, 1, 4, 4, tab) != 0
    # null row pointers inside a valid call are rejected before any launch
    with pytest.raises(_native.StError):
        hip_ctx._check(L.st_box_blur_u8c3_batch(hip_ctx._h, tab, 1, 4, 4, 3, tab))
    assert L.st_cvt_color_out_channels(6, 3) == 1 and L.st_cvt_color_out_channels(6, 1) == -1
    assert L.st_cvt_color_out_channels(12345, 3) == -1
