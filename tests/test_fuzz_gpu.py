"""Seeded differential fuzzing of every op against the CPU oracle through the C ABI: random small and
ragged shapes, random parameters.  Integer ops must agree bit for bit, flows within the stated
tolerance."""
import numpy as np
import pytest
import torch

import oracle
from scannertools_amd._native import COLOR_CODES
from util import assert_flow_close, cvt_source, smooth_texture

pytestmark = pytest.mark.gpu
FLOW_TIERS = {1: 0, 2: 0, 3: 0}   # how many flow fields passed under which tier of util.assert_flow_close (campaign statistics)


def _cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_integer_ops(hip_ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(8):
        h, w, n = int(rng.integers(1, 98)), int(rng.integers(1, 132)), int(rng.integers(1, 4))
        frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        if rng.random() < 0.3:                                   # low-entropy frames: heavy bin collisions
            frames = (frames // 64 * 64).astype(np.uint8)
        fr = _cu(frames)
        bins = int(rng.integers(1, 257))
        got = hip_ctx.histogram(fr, bins).cpu().numpy()
        for i in range(n):
            np.testing.assert_array_equal(got[i], oracle.hist_u8c3(frames[i], bins), err_msg="hist %dx%d bins %d" % (h, w, bins))
        k = int(rng.integers(1, 12))
        got = hip_ctx.box_blur(fr, k).cpu().numpy()
        for i in range(n):
            np.testing.assert_array_equal(got[i], oracle.box_blur(frames[i], k), err_msg="blur %dx%d k %d" % (h, w, k))
        dh, dw, interp = int(rng.integers(1, 120)), int(rng.integers(1, 150)), int(rng.integers(0, 5))
        got = hip_ctx.resize(fr, dw, dh, interp).cpu().numpy()
        for i in range(n):
            np.testing.assert_array_equal(got[i], oracle.resize_u8(frames[i], dw, dh, interp),
                                          err_msg="resize %dx%d -> %dx%d interp %d" % (h, w, dh, dw, interp))
        name = list(COLOR_CODES)[int(rng.integers(0, len(COLOR_CODES)))]
        src = np.stack([cvt_source(rng, COLOR_CODES[name], h, w) for _ in range(n)])
        got = hip_ctx.cvt_color(_cu(src), name).cpu().numpy()
        for i in range(n):
            np.testing.assert_array_equal(got[i], oracle.cvt_color(src[i], COLOR_CODES[name]), err_msg="%s %dx%d" % (name, h, w))
        flows = (rng.standard_normal((n, h, w, 2)) * float(rng.choice([0.3, 3.0, 40.0]))).astype(np.float32)
        got = hip_ctx.flow_histogram(_cu(flows)).cpu().numpy()
        for i in range(n):
            np.testing.assert_array_equal(got[i], oracle.flow_hist(flows[i]), err_msg="flow_hist %dx%d" % (h, w))
        got = hip_ctx.draw_flow(fr, _cu(flows)).cpu().numpy()
        for i in range(n):
            np.testing.assert_array_equal(got[i], oracle.draw_flow(frames[i], flows[i]), err_msg="draw_flow %dx%d" % (h, w))


# seeds >= 100: the pairs of the round-3 campaigns (300 + 1 500 seeds) that need tiers 2 / 3 of util.assert_flow_close.
# EXPECTED_TIERS pins what each seed is allowed: the tier of every one of its flow fields, in order (the results are the
# same bits under every scheduling mode, so one list per seed).  A field that needs a HIGHER tier than recorded fails --
# seeds 0, 1, 2 are plain-bounds seeds and must stay so -- and a field that needs a lower one is reported, so that the list
# is tightened rather than left to rot.
EXPECTED_TIERS = {}
EXPECTED_TIERS_FILE = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "flow_fuzz_tiers.json")


def _expected_tiers():
    if not EXPECTED_TIERS:
        import json
        EXPECTED_TIERS.update({int(k): v for k, v in json.load(open(EXPECTED_TIERS_FILE)).items()})
    return EXPECTED_TIERS


@pytest.mark.parametrize("seed", [0, 1, 2, 103, 121, 268, 520, 976, 1420, 2876, 4079])
def test_fuzz_optical_flow(flow_ctx, seed):
    """Random frame sizes (including ones smaller than the window and ones that change the number of
    pyramid levels) and random pair lists; every flow field against the oracle, under every
    scheduling mode of the flow iteration, each at the tolerance tier recorded for it."""
    hip_ctx = flow_ctx
    rng = np.random.default_rng(77 + seed)
    tiers = []
    for _ in range(4):
        h, w = int(rng.integers(2, 150)), int(rng.integers(2, 200))
        nf = int(rng.integers(2, 5))
        base = np.stack([smooth_texture(int(rng.integers(1 << 30)), h + 8, w + 8) for _ in range(3)], -1)
        frames = np.stack([base[dy:dy + h, dx:dx + w] for dy, dx in rng.integers(0, 9, (nf, 2))]).astype(np.uint8)
        pairs = [(int(a), int(b)) for a, b in rng.integers(0, nf, (int(rng.integers(1, 5)), 2))]
        got = hip_ctx.optical_flow(_cu(frames), pairs=pairs).cpu().numpy()
        for i, (a, b) in enumerate(pairs):
            ref = oracle.optical_flow_rgb(frames[a], frames[b])
            tier = assert_flow_close(got[i], ref, frames[a], frames[b], (h, w, a, b))
            FLOW_TIERS[tier] += 1
            tiers.append(tier)
    import os
    if os.environ.get("ST_RECORD_FLOW_TIERS"):   # (re)generate tests/golden/flow_fuzz_tiers.json: see scripts/record_flow_tiers.sh
        print("FLOW_TIERS_RECORD %d %s" % (seed, tiers))
        return
    want = _expected_tiers()[seed]
    assert len(tiers) == len(want), (seed, tiers, want)
    worse = [(i, t, e) for i, (t, e) in enumerate(zip(tiers, want)) if t > e]
    assert not worse, "seed %d: flow fields needing a higher tolerance tier than recorded (field, needed, recorded): %s" % (seed, worse)
    if seed in (0, 1, 2):
        assert max(tiers) == 1, (seed, tiers)


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_pose_ops(hip_ctx, seed):
    """CPM2Input at random frame sizes and scales; the `resize` layer at random (also non-integer, also shrinking)
    ratios with random channel maps; the `nms` layer on random peak populations; the limb scores on the result."""
    rng = np.random.default_rng(5000 + seed)
    for _ in range(6):
        h, w, n = int(rng.integers(8, 200)), int(rng.integers(8, 260)), int(rng.integers(1, 4))
        scale = float(rng.uniform(0.2, 1.6))
        if int(h * np.float32(scale)) < 1 or int(w * np.float32(scale)) < 1:
            continue
        frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        got = hip_ctx.cpm2_input(_cu(frames), scale).cpu().numpy()
        for i in range(n):
            np.testing.assert_array_equal(got[i], oracle.cpm2_input(frames[i], scale), err_msg="cpm2_input %dx%d scale %r" % (h, w, scale))
        sh, sw, C = int(rng.integers(1, 40)), int(rng.integers(1, 50)), int(rng.integers(1, 24))
        th, tw = int(rng.integers(1, 180)), int(rng.integers(1, 300))
        maps = rng.standard_normal((n, sh, sw, C)).astype(np.float32)
        chan = [int(c) for c in rng.integers(0, C, int(rng.integers(1, 20)))] if rng.random() < 0.5 else None
        got = hip_ctx.cpm2_resize_maps(_cu(maps), th, tw, chan_map=chan).cpu().numpy()
        sel = chan if chan is not None else list(range(C))
        for i in range(n):
            np.testing.assert_array_equal(got[i], oracle.cpm2_resize_maps(np.ascontiguousarray(maps[i].transpose(2, 0, 1)[sel]), th, tw),
                                          err_msg="resize_maps %dx%d -> %dx%d" % (sh, sw, th, tw))
        H, W, mp = int(rng.integers(1, 90)), int(rng.integers(1, 140)), int(rng.integers(1, 20))
        hm = (rng.random((n, 57, H, W)) * 0.06).astype(np.float32)
        mask = rng.random(hm.shape) < float(rng.choice([0.001, 0.02, 0.3]))
        hm[mask] = rng.random(int(mask.sum())).astype(np.float32)
        thr = float(rng.choice([0.05, 0.5]))
        joints = hip_ctx.cpm2_nms(_cu(hm), parts=18, max_peaks=mp, threshold=thr)
        jn = joints.cpu().numpy()
        for i in range(n):
            np.testing.assert_array_equal(jn[i], oracle.cpm2_nms(hm[i], 18, mp, thr), err_msg="nms %dx%d max %d" % (H, W, mp))
        sc = hip_ctx.cpm2_limb_scores(_cu(hm), joints).cpu().numpy()
        for i in range(n):
            np.testing.assert_array_equal(sc[i], oracle.cpm2_limb_scores(hm[i], jn[i]), err_msg="limb scores %dx%d" % (H, W))
