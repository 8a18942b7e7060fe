"""GPU parity: Farneback OpticalFlow (HIP, through the C ABI) vs the CPU oracle.

Stage tests compare each kernel with the oracle's restatement of the same OpenCV routine on
identical inputs.  The library is built with -ffp-contract=off and follows the scalar operand
order, so every stage whose arithmetic is per-pixel (gray, pyramid, polynomial expansion,
UpdateMatrices) must be BIT-EXACT.  The box filter re-anchors its running double sums every 32
rows / 8 columns (the reference accumulates float-rounded differences from the top / left of the
image), so blur-dependent outputs are compared within a tolerance:
  stage   UpdateFlow_Blur : |d flow| <= 1e-4 px, |d M'| <= 1e-4 * max|M'|
  end-to-end flow         : relative L2 <= 1e-4 and max-abs <= 5e-3 px on textured pairs
(north-star bound: relative L2 <= 1e-3, max-abs <= 1e-2 px).

The flow iteration has two kernels (marching k_flow_iter3 and k_flow_iter_tile) and a launch-size
rule that picks between them; the ``flow_ctx`` fixture runs a test under every scheduling mode
(conftest.FLOW_MODES), so each kernel instance -- zero / field / coarse / half-height coarse flow
source, marching and tile -- is compared with the oracle, and ``test_schedules_agree_bitwise``
checks that the modes agree to the last bit.
"""
import numpy as np
import pytest
import torch

import oracle
from scannertools_amd.hip import default_params
from util import interleaved5, planar5, random_frames, smooth_texture, texture_stream, translated_rgb_pair

pytestmark = pytest.mark.gpu

REL_L2_TOL = 1e-4
MAX_ABS_TOL = 5e-3

SIZES = [(48, 64), (97, 131), (240, 320), (203, 317)]


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


# ---------------------------------------------------------------- A2 gray
@pytest.mark.parametrize("bits", [14, 15])
def test_gray_bit_exact(hip_ctx, bits):
    f = random_frames(bits, 1, 101, 173)[0]
    got = hip_ctx.gray(cu(f), bits).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.gray_u8(f, bits))


def test_gray_known_answers(hip_ctx):
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0]]], np.uint8)
    for bits, cb, cg, cr in ((15, 3735, 19235, 9798), (14, 1868, 9617, 4899)):
        got = hip_ctx.gray(cu(px), bits).cpu().numpy()[0]
        rnd = 1 << (bits - 1)
        # channel-swap quirk: the R byte gets the B coefficient (0.114)
        assert got[0] == (255 * cb + rnd) >> bits
        assert got[1] == (255 * cg + rnd) >> bits
        assert got[2] == (255 * cr + rnd) >> bits
        assert got[3] == 255 and got[4] == 0


# ---------------------------------------------------------------- pyramid
# (240, 426): the legacy flow-histogram pipeline's frame (old/histograms.py:66-67) -- three levels, rows that start at any
# byte; widths of every residue mod 4 take the streaming level-0 kernel, the last of them with a one-column tail group
@pytest.mark.parametrize("h,w", [(240, 320), (203, 317), (270, 480), (135, 241), (240, 426), (61, 1027), (40, 1030), (33, 9)])
def test_pyramid_levels_bit_exact(hip_ctx, h, w):
    gray = (smooth_texture(h + w, h, w) + np.random.default_rng(1).integers(0, 8, (h, w))).clip(0, 255).astype(np.uint8)
    p = default_params()
    levels = oracle.fb_levels(h, w)
    for k in range(levels + 1):
        got = hip_ctx.pyr_image(cu(gray), k, p).cpu().numpy()
        ref = oracle.fb_pyr_image(gray, k)
        assert got.shape == ref.shape
        np.testing.assert_array_equal(got, ref, err_msg="level %d" % k)


@pytest.mark.parametrize("h,w", [(256, 256), (264, 1032), (512, 2056), (328, 1024), (2160, 3840)])
def test_pyramid_one_pass_kernel_bit_exact(hip_ctx, h, w):
    """Frames whose sides are multiples of 8 (>= 256) take the one-pass four-level kernel: single
    strips, a second strip of 8 columns, three strips, several vertical segments, 4K."""
    assert oracle.fb_levels(h, w) == 3
    gray = np.random.default_rng(h + w).integers(0, 256, (h, w), dtype=np.uint8)
    for k in range(4):
        got = hip_ctx.pyr_image(cu(gray), k).cpu().numpy()
        assert got.shape == (h >> k, w >> k)
        np.testing.assert_array_equal(got, oracle.fb_pyr_image(gray, k), err_msg="level %d" % k)


def test_pyramid_1080p_geometry_and_parity(hip_ctx):
    h, w = 1080, 1920
    assert oracle.fb_levels(h, w) == 3
    gray = np.random.default_rng(0).integers(0, 256, (h, w), dtype=np.uint8)
    for k, shape in enumerate([(1080, 1920), (540, 960), (270, 480), (135, 240)]):
        got = hip_ctx.pyr_image(cu(gray), k).cpu().numpy()
        assert got.shape == shape
        np.testing.assert_array_equal(got, oracle.fb_pyr_image(gray, k))


# ---------------------------------------------------------------- A4 polyexp
@pytest.mark.parametrize("h,w", SIZES + [(12, 300), (300, 9)])
def test_polyexp_bit_exact(hip_ctx, h, w):
    I = (smooth_texture(h * w, h, w) * 1.0).astype(np.float32)
    got = hip_ctx.polyexp(cu(I)).cpu().numpy()
    ref = oracle.polyexp(I)
    np.testing.assert_array_equal(got, ref)


def test_polyexp_exact_quadratic(hip_ctx):
    h, w = 64, 80
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    a, bx, by, cxx, cyy, cxy = 3.0, 0.5, -0.25, 0.01, 0.02, -0.015
    I = (a + bx * (x - 40) + by * (y - 30) + cxx * (x - 40) ** 2 + cyy * (y - 30) ** 2 + cxy * (x - 40) * (y - 30))
    R = hip_ctx.polyexp(cu(I.astype(np.float32))).cpu().numpy()
    yy, xx = 20, 25
    exp = [by + 2 * cyy * (yy - 30) + cxy * (xx - 40), bx + 2 * cxx * (xx - 40) + cxy * (yy - 30), cyy, cxx, cxy]
    np.testing.assert_allclose(R[yy, xx, :], exp, atol=2e-6)


def test_polyexp_n7(hip_ctx):
    I = smooth_texture(3, 90, 120).astype(np.float32)
    got = hip_ctx.polyexp(cu(I), 7, 1.5).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.polyexp(I, 7, 1.5))


# ---------------------------------------------------------------- A5 UpdateMatrices
def _expansions(seed, h, w, tx=2, ty=-1):
    f0, f1 = translated_rgb_pair(seed, h, w, tx, ty)
    g0, g1 = oracle.gray_u8(f0), oracle.gray_u8(f1)
    R0 = oracle.polyexp(g0.astype(np.float32))
    R1 = oracle.polyexp(g1.astype(np.float32))
    return R0, R1


@pytest.mark.parametrize("h,w", SIZES)
def test_update_matrices_bit_exact(hip_ctx, h, w):
    R0, R1 = _expansions(h, h, w)
    rng = np.random.default_rng(w)
    flow = (rng.standard_normal((h, w, 2)) * 3).astype(np.float32)
    flow[0, 0] = (-50, -50)      # far outside -> else-branch
    flow[-1, -1] = (0, 0)        # last row/col quirk
    got = hip_ctx.update_matrices(cu(R0), cu(R1), flow=cu(flow)).cpu().numpy()
    np.testing.assert_array_equal(got, planar5(oracle.update_matrices(R0, R1, flow)))
    # zero flow
    got0 = hip_ctx.update_matrices(cu(R0), cu(R1)).cpu().numpy()
    np.testing.assert_array_equal(got0, planar5(oracle.update_matrices(R0, R1, np.zeros((h, w, 2), np.float32))))


@pytest.mark.parametrize("h,w,ch,cw", [(96, 128, 48, 64), (97, 131, 48, 66), (203, 317, 102, 158)])
def test_update_matrices_with_flow_upsample_bit_exact(hip_ctx, h, w, ch, cw):
    R0, R1 = _expansions(h + 1, h, w)
    coarse = (np.random.default_rng(ch).standard_normal((ch, cw, 2)) * 2).astype(np.float32)
    up = oracle.resize_linear(coarse, h, w) * np.float32(2.0)
    got = hip_ctx.update_matrices(cu(R0), cu(R1), coarse_flow=cu(coarse), pyr_scale=0.5).cpu().numpy()
    np.testing.assert_array_equal(got, planar5(oracle.update_matrices(R0, R1, up)))


def test_update_matrices_identical_frames_quirk(hip_ctx):
    """R0 == R1 and zero flow: h1 = h2 = 0 inside, non-zero on the last row / column (A5 quirk)."""
    h, w = 60, 70
    R0, _ = _expansions(9, h, w)
    M = hip_ctx.update_matrices(cu(R0), cu(R0)).cpu().numpy()
    assert np.abs(M[3:, :-1, :-1]).max() == 0
    assert np.abs(M[3:, -1, :]).max() > 0 and np.abs(M[3:, :, -1]).max() > 0


# ---------------------------------------------------------------- A6 UpdateFlow_Blur
@pytest.mark.parametrize("h,w", SIZES + [(20, 500), (600, 24)])
@pytest.mark.parametrize("update", [True, False])
def test_update_flow_blur_parity(hip_ctx, h, w, update):
    R0, R1 = _expansions(h + 2, h, w)
    M = oracle.update_matrices(R0, R1, np.zeros((h, w, 2), np.float32))
    ref_flow, ref_M = oracle.update_flow_blur(R0, R1, M, 15, update)
    flow, Mn = hip_ctx.update_flow_blur(cu(R0), cu(R1), cu(planar5(M)), 15, update)
    flow = flow.cpu().numpy()
    assert np.abs(flow - ref_flow).max() <= 1e-4
    if update:
        Mn = interleaved5(Mn.cpu().numpy())
        assert np.abs(Mn - ref_M).max() <= 1e-4 * np.abs(ref_M).max()
    else:
        assert Mn is None


def test_update_flow_blur_other_window(hip_ctx):
    h, w = 80, 90
    R0, R1 = _expansions(4, h, w)
    M = oracle.update_matrices(R0, R1, np.zeros((h, w, 2), np.float32))
    for bs in (5, 21):
        ref_flow, _ = oracle.update_flow_blur(R0, R1, M, bs, False)
        flow, _ = hip_ctx.update_flow_blur(None, None, cu(planar5(M)), bs, False)
        assert np.abs(flow.cpu().numpy() - ref_flow).max() <= 1e-4


# ---------------------------------------------------------------- fused iteration (production kernel)
ITER_SIZES = SIZES + [(20, 500), (600, 24), (135, 240), (270, 480), (540, 960)]


def _iteration_cases(h, w):
    """Inputs of the three flow sources of one iteration at (h, w) + the oracle's results."""
    R0, R1 = _expansions(h + 3, h, w)
    rng = np.random.default_rng(h)
    cases = {}
    M = oracle.update_matrices(R0, R1, np.zeros((h, w, 2), np.float32))
    cases["zero"] = (dict(), oracle.update_flow_blur(R0, R1, M, 15, False)[0])
    fin = (rng.standard_normal((h, w, 2)) * 2).astype(np.float32)
    M = oracle.update_matrices(R0, R1, fin)
    cases["field"] = (dict(flow_in=fin), oracle.update_flow_blur(R0, R1, M, 15, False)[0])
    # exactly half as tall when h is even (FLOW_COARSE2 instance), the generic instance otherwise
    ch, cw = (h + 1) // 2, (w + 1) // 2
    coarse = (rng.standard_normal((ch, cw, 2)) * 2).astype(np.float32)
    up = oracle.resize_linear(coarse, h, w) * np.float32(2.0)
    M = oracle.update_matrices(R0, R1, up)
    cases["coarse"] = (dict(coarse_flow=coarse, pyr_scale=0.5), oracle.update_flow_blur(R0, R1, M, 15, False)[0])
    return R0, R1, cases


@pytest.mark.parametrize("h,w", ITER_SIZES)
def test_flow_iteration_parity(mode_ctxs, h, w):
    """One fused iteration == UpdateMatrices followed by UpdateFlow_Blur, for the three flow sources,
    under the marching kernel (k_flow_iter3), the tile kernel (k_flow_iter_tile) and the role-split marching kernel
    (k_flow_iter_roles, 4 and 5 column waves) alike; and all of them agree bit for bit."""
    R0, R1, cases = _iteration_cases(h, w)
    r0, r1 = cu(R0), cu(R1)
    for name, (kw, ref) in cases.items():
        outs = {}
        for mode in ("march", "tile", "roles4", "roles5"):
            got = mode_ctxs[mode].flow_iteration(r0, r1, **{k: (cu(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()})
            outs[mode] = got.cpu().numpy()
            assert np.abs(outs[mode] - ref).max() <= 1e-4, (name, mode, np.abs(outs[mode] - ref).max())
        for mode in ("tile", "roles4", "roles5"):
            np.testing.assert_array_equal(outs["march"], outs[mode], err_msg="%s %s" % (name, mode))


# ---------------------------------------------------------------- A3 end to end
def _check_flow(got, ref):
    assert got.shape == ref.shape and got.dtype == np.float32
    assert rel_l2(got, ref) <= REL_L2_TOL, rel_l2(got, ref)
    assert np.abs(got - ref).max() <= MAX_ABS_TOL, np.abs(got - ref).max()


@pytest.mark.parametrize("h,w", [(240, 320), (203, 317), (480, 640), (48, 64)])
def test_flow_matches_oracle(flow_ctx, h, w):
    hip_ctx = flow_ctx
    f0, f1 = translated_rgb_pair(h, h, w, 3, -2)
    got = hip_ctx.optical_flow(cu(np.stack([f0, f1]))).cpu().numpy()
    assert got.shape == (1, h, w, 2)
    _check_flow(got[0], oracle.optical_flow_rgb(f0, f1))


def test_flow_recovers_translation(hip_ctx):
    h, w, tx, ty = 240, 320, 3, -2
    f0, f1 = translated_rgb_pair(11, h, w, tx, ty)
    fl = hip_ctx.optical_flow(cu(np.stack([f0, f1]))).cpu().numpy()[0]
    inner = fl[40:-40, 40:-40]
    assert abs(np.median(inner[..., 0]) - tx) < 0.01 and abs(np.median(inner[..., 1]) - ty) < 0.01
    assert np.abs(inner - [tx, ty]).mean() < 0.01


def test_flow_identical_frames(hip_ctx):
    f0, _ = translated_rgb_pair(5, 240, 320, 0, 0)
    fl = hip_ctx.optical_flow(cu(np.stack([f0, f0]))).cpu().numpy()[0]
    assert np.abs(fl[:100, :150]).max() < 0.05          # far from the right/bottom border quirk (u8 quantisation noise only)
    _check_flow(fl, oracle.optical_flow_rgb(f0, f0))


def test_flow_stream_batch_pairs_and_direction(flow_ctx):
    """Batched stencil {0,1} over a stream, arbitrary pairs, CPU-kernel direction."""
    hip_ctx = flow_ctx
    h, w = 120, 160
    frames, _ = texture_stream(2, 6, h, w)
    d = cu(frames)
    got = hip_ctx.optical_flow(d).cpu().numpy()
    assert got.shape == (5, h, w, 2)
    for i in range(5):
        _check_flow(got[i], oracle.optical_flow_rgb(frames[i], frames[i + 1]))
    # arbitrary, repeated and reversed pairs; list-of-buffers input
    pairs = [(4, 1), (1, 4), (0, 0), (2, 5)]
    got2 = hip_ctx.optical_flow([d[i] for i in range(6)], pairs).cpu().numpy()
    for k, (a, b) in enumerate(pairs):
        _check_flow(got2[k], oracle.optical_flow_rgb(frames[a], frames[b]))
    assert np.abs(got2[0] - got2[1]).max() > 0.1  # direction matters


@pytest.mark.parametrize("h,w,n", [(135, 240, 4), (256, 320, 6), (1080, 1920, 4)])
def test_flow_batch_equals_single(hip_ctx, h, w, n):
    """A pair's flow must not depend on how a stream is cut into calls (Scanner's work packets are
    ragged and the reference is deterministic per pair): whole batch == one pair per call == an
    uneven split, bit for bit, in the default configuration -- although the three use different
    kernels and segment heights."""
    if h == 1080:
        g = torch.Generator(device="cuda").manual_seed(3)
        low = torch.rand((1, 3, h // 8 + 8, w // 8 + 8), device="cuda", generator=g)
        tex = torch.nn.functional.interpolate(low, size=(h + 32, w + 32), mode="bicubic", align_corners=False)[0]
        tex = ((tex - tex.amin()) / (tex.amax() - tex.amin()) * 255).permute(1, 2, 0)
        d = torch.stack([tex[16 + i:16 + i + h, 16 - 2 * i:16 - 2 * i + w].to(torch.uint8) for i in range(n)]).contiguous()
    else:
        d = cu(texture_stream(3, n, h, w)[0])
    batch = hip_ctx.optical_flow(d).cpu().numpy()
    single = torch.stack([hip_ctx.optical_flow(d[i:i + 2])[0] for i in range(n - 1)]).cpu().numpy()
    np.testing.assert_array_equal(batch, single)
    split = torch.cat([hip_ctx.optical_flow(d[:3]), hip_ctx.optical_flow(d[2:])]).cpu().numpy()
    np.testing.assert_array_equal(batch, split)


def test_schedules_agree_bitwise(mode_ctxs):
    """Every scheduling mode (kernel choice, pairs per workgroup) gives the same bits end to end."""
    for (h, w, n) in ((135, 240, 3), (203, 317, 4), (544, 960, 5), (256, 1032, 2)):
        d = cu(texture_stream(h, n, h, w)[0])
        ref = mode_ctxs["march"].optical_flow(d).cpu().numpy()
        for mode in ("default", "tile", "foldgray", "roles4", "roles5"):
            np.testing.assert_array_equal(mode_ctxs[mode].optical_flow(d).cpu().numpy(), ref, err_msg="%s %dx%d" % (mode, h, w))


def test_flow_other_params(hip_ctx):
    h, w = 160, 200
    f0, f1 = translated_rgb_pair(8, h, w, -2, 1)
    for kw in (dict(num_levels=1), dict(num_levels=5), dict(win_size=9, num_iters=2), dict(gray_bits=14),
               dict(pyr_scale=0.7, num_levels=2), dict(poly_n=7, poly_sigma=1.5)):
        ref = oracle.optical_flow_rgb(f0, f1, oracle.default_params(**kw))
        got = hip_ctx.optical_flow(cu(np.stack([f0, f1])), params=default_params(**kw)).cpu().numpy()[0]
        _check_flow(got, ref)


def test_flow_1080p_pair(flow_ctx):
    hip_ctx = flow_ctx
    h, w = 1080, 1920
    f0, f1 = translated_rgb_pair(21, h, w, 4, 3)
    got = hip_ctx.optical_flow(cu(np.stack([f0, f1]))).cpu().numpy()[0]
    _check_flow(got, oracle.optical_flow_rgb(f0, f1))
    inner = got[100:-100, 100:-100]
    assert abs(np.median(inner[..., 0]) - 4) < 0.05 and abs(np.median(inner[..., 1]) - 3) < 0.05


def _torch_stream(n, h, w, seed, step=2):
    """n frames of a smooth texture under a steady translation, generated on the GPU (big sizes)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    m = step * n + 8
    low = torch.rand((1, 3, (h + 2 * m) // 8 + 2, (w + 2 * m) // 8 + 2), device="cuda", generator=g)
    tex = torch.nn.functional.interpolate(low, size=(h + 2 * m, w + 2 * m), mode="bicubic", align_corners=False)[0]
    tex = ((tex - tex.amin()) / (tex.amax() - tex.amin()) * 235 + 10).permute(1, 2, 0)
    fr = [tex[m + i:m + i + h, m - step * i:m - step * i + w] + torch.randint(-2, 3, (h, w, 3), device="cuda", generator=g)
          for i in range(n)]
    return torch.stack([f.clamp(0, 255).to(torch.uint8) for f in fr]).contiguous()


def test_flow_1080p_batch_launch_geometry(mode_ctxs):
    """24 pairs of 1080p in one call: every pyramid level takes the marching kernel (all four flow
    sources of k_flow_iter3, multi-round segment geometry as in the 256-pair benchmark call); a
    sample of the pairs against the oracle, all of them against the schedule that folds the luma
    conversion into the pyramid, and a sub-batch against the full batch."""
    h, w, n = 1080, 1920, 25
    d = _torch_stream(n, h, w, 9)
    got = mode_ctxs["default"].optical_flow(d)
    np_frames = d.cpu().numpy()
    for i in (0, 7, 16, 23):
        _check_flow(got[i].cpu().numpy(), oracle.optical_flow_rgb(np_frames[i], np_frames[i + 1]))
    assert torch.equal(got, mode_ctxs["foldgray"].optical_flow(d))
    assert torch.equal(mode_ctxs["default"].optical_flow(d[:4]), got[:3])


@pytest.mark.parametrize("h,w,n", [(256, 256, 18), (264, 1032, 18), (328, 1024, 20), (1080, 1920, 19), (2160, 3840, 18)])
def test_polyexp_from_gray_frames_bit_exact(mode_ctxs, h, w, n):
    """Calls above 16 pairs on one-pass-pyramid geometries: the level-0 expansion reads the GRAY frames and evaluates the
    3 x 3 blur itself (k_polyexp_u8; level 0 is left out of k_pyr_roles).  Every intermediate of that blur is exact in
    float, so the flows must equal the float-source schedule's bit for bit: single strips, a strip of 8 columns, several
    segments, 1080p, 4K; random bytes (every edge weight and reflected row matters) and a textured stream."""
    g = torch.Generator(device="cuda").manual_seed(h + w)
    noise = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    for d in (noise, _torch_stream(n, h, w, h)):
        a = mode_ctxs["polyu8"].optical_flow(d)
        b = mode_ctxs["polyf32"].optical_flow(d)
        assert torch.equal(a, b)
        del a, b
    if h <= 1080:
        fr = _torch_stream(n, h, w, h)
        got = mode_ctxs["polyu8"].optical_flow(fr)[n // 2].cpu().numpy()
        f = fr.cpu().numpy()
        _check_flow(got, oracle.optical_flow_rgb(f[n // 2], f[n // 2 + 1]))


@pytest.mark.parametrize("h,w", [(1, 1), (1, 40), (40, 1), (2, 2), (3, 5), (31, 33)])
def test_flow_tiny_frames(hip_ctx, h, w):
    """Degenerate geometries (single level, clamps everywhere) still match the oracle."""
    f = random_frames(h * 7 + w, 2, h, w)
    got = hip_ctx.optical_flow(cu(f)).cpu().numpy()[0]
    ref = oracle.optical_flow_rgb(f[0], f[1])
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max())


def test_flow_rejects_unsupported(hip_ctx):
    from scannertools_amd.hip import StError
    f = torch.zeros((2, 64, 64, 3), dtype=torch.uint8, device="cuda")
    for kw in (dict(flags=256), dict(fast_pyramids=1), dict(poly_n=6), dict(win_size=14), dict(pyr_scale=1.5)):
        with pytest.raises(StError):
            hip_ctx.optical_flow(f, params=default_params(**kw))
    with pytest.raises(StError):
        hip_ctx.optical_flow(f, pairs=[(0, 2)])
    assert tuple(hip_ctx.optical_flow(f, pairs=np.zeros((0, 2), np.int32)).shape) == (0, 64, 64, 2)


# ---------------------------------------------------------------- BASELINE config 4 size (4K)
def test_flow_4k_properties(hip_ctx):
    """3840x2160 (config 4): level geometry, planted translation, batch == single, zero on identical."""
    from scannertools_amd.hip import fb_level_geom, fb_levels
    h, w = 2160, 3840
    assert fb_levels(h, w) == 3 and fb_level_geom(h, w, 3)[:2] == (270, 480)
    g = torch.Generator(device="cuda").manual_seed(5)
    low = torch.rand((1, 3, h // 8 + 12, w // 8 + 12), device="cuda", generator=g)
    tex = torch.nn.functional.interpolate(low, size=(h + 64, w + 64), mode="bicubic", align_corners=False)[0]
    tex = ((tex - tex.amin()) / (tex.amax() - tex.amin()) * 255).permute(1, 2, 0)
    f0 = tex[32:32 + h, 32:32 + w].to(torch.uint8).contiguous()
    f1 = tex[32 - 2:32 - 2 + h, 32 - 5:32 - 5 + w].to(torch.uint8).contiguous()      # next(x+5, y+2) = prev(x, y)
    fr = torch.stack([f0, f1, f0])
    fl = hip_ctx.optical_flow(fr, pairs=[(0, 1), (1, 2), (0, 2)])
    inner = fl[0, 200:-200, 200:-200]
    assert abs(float(inner[..., 0].median()) - 5) < 0.05 and abs(float(inner[..., 1].median()) - 2) < 0.05
    inner = fl[1, 200:-200, 200:-200]
    assert abs(float(inner[..., 0].median()) + 5) < 0.05 and abs(float(inner[..., 1].median()) + 2) < 0.05
    assert float(fl[2, :1500, :3000].abs().max()) < 0.05                              # identical frames
    single = hip_ctx.optical_flow(fr[:2])
    assert torch.equal(fl[0], single[0])


def test_flow_4k_pair_matches_oracle(hip_ctx):
    """Config 4 resolution against the oracle (one pair; the oracle needs a few seconds at 4K)."""
    h, w = 2160, 3840
    d = _torch_stream(2, h, w, 17, step=3)
    got = hip_ctx.optical_flow(d).cpu().numpy()[0]
    fr = d.cpu().numpy()
    _check_flow(got, oracle.optical_flow_rgb(fr[0], fr[1]))


def test_flow_in_passes_under_a_scratch_cap():
    """A small workspace limit makes st_farneback_pairs split the batch into passes (frames shared
    by consecutive pairs are then expanded once per pass), for the one-pass pyramid geometry
    (256x320) and for a generic one (203x317).  The passes reproduce the single call bit for bit
    (the launches differ in size, hence in kernel and segment choice; the results must not).  A cap
    below what one pair needs is an error, not a fallback."""
    from conftest import make_mode_ctx
    from scannertools_amd._native import StError
    for (h, w) in ((256, 320), (203, 317)):
        frames, _ = texture_stream(h, 10, h, w)
        fr = torch.from_numpy(frames).cuda()
        pairs = [(i, i + 1) for i in range(9)] + [(4, 2), (7, 7)]
        per_pair = 4 * h * w * (5 * 2 * 1.4 + 2 * 2 + 2)      # rough: two expansions + flow buffers
        for mode in ("default", "march"):
            with make_mode_ctx(mode) as big:
                ref = big.optical_flow(fr, pairs=pairs).cpu().numpy()
            with make_mode_ctx(mode, workspace_limit=int(3.5 * per_pair)) as small:
                got = small.optical_flow(fr, pairs=pairs).cpu().numpy()
            np.testing.assert_array_equal(got, ref)
        with make_mode_ctx("default", workspace_limit=int(0.2 * per_pair)) as tiny:
            with pytest.raises(StError):
                tiny.optical_flow(fr, pairs=pairs)
