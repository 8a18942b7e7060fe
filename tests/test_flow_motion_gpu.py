"""GPU parity of the OpticalFlow op on motion that is NOT a whole-pixel shift.

The reference op runs on decoded video (scannertools/tests/test_all.py:162-177 runs OpticalFlow over a range of a real
clip; scannertools_infra/tests.py:17-86 downloads it): sub-pixel translation, zoom, rotation, an occlusion edge, a large
jump, a static scene, pure noise.  Every other end-to-end flow test of this suite shifts one texture by whole pixels;
here each kind of tests/util.py: motion_pair goes through the C ABI
  * at 480x640 under EVERY scheduling mode of the flow iteration (conftest.FLOW_MODES) and
  * at 1080x1920 once (the default schedule),
against the oracle through util.assert_flow_close, each case pinned to the tolerance tier recorded for it
(tests/golden/flow_motion_tiers.json; regenerate on a GPU box with ST_RECORD_FLOW_TIERS=1, see the bottom of this file).
The schedules must also agree with each other to the last bit on every kind (a pair's flow is a function of the pair).
The same pairs are in tests/golden/make_opencv_golden.py, so the one run elsewhere that pins the oracle pins these too.
"""
import functools
import json
import os

import numpy as np
import pytest
import torch

import oracle
from util import MOTION_KINDS, assert_flow_close, motion_pair

pytestmark = pytest.mark.gpu

TIERS_FILE = os.path.join(os.path.dirname(__file__), "golden", "flow_motion_tiers.json")
MOTION_SEED = 3


@functools.lru_cache(maxsize=None)
def _case(kind, h, w):
    a, b = motion_pair(kind, MOTION_SEED, h, w)
    return a, b, oracle.optical_flow_rgb(a, b)


def _tiers():
    return json.load(open(TIERS_FILE))


def _run(ctx, kind, h, w):
    a, b, ref = _case(kind, h, w)
    got = ctx.optical_flow(torch.from_numpy(np.stack([a, b])).cuda()).cpu().numpy()[0]
    tier = assert_flow_close(got, ref, a, b, (kind, h, w))
    if os.environ.get("ST_RECORD_FLOW_TIERS"):
        d = np.abs(got - ref)
        print("MOTION_TIER_RECORD %s %dx%d %d rel_l2 %.3g max_abs %.3g" %
              (kind, h, w, tier, np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30), d.max()))
        return got
    want = _tiers()["%s_%dx%d" % (kind, h, w)]
    assert tier <= want, "%s %dx%d needs tolerance tier %d, recorded %d" % (kind, h, w, tier, want)
    return got


@pytest.mark.parametrize("kind", MOTION_KINDS)
def test_flow_motion_kinds_every_schedule(flow_ctx, kind):
    _run(flow_ctx, kind, 480, 640)


@pytest.mark.parametrize("kind", MOTION_KINDS)
def test_flow_motion_kinds_schedules_agree_bitwise(mode_ctxs, kind):
    a, b, _ = _case(kind, 480, 640)
    d = torch.from_numpy(np.stack([a, b])).cuda()
    ref = mode_ctxs["default"].optical_flow(d)
    for mode, ctx in mode_ctxs.items():
        assert torch.equal(ctx.optical_flow(d), ref), (kind, mode)


@pytest.mark.parametrize("kind", MOTION_KINDS)
def test_flow_motion_kinds_1080p(hip_ctx, kind):
    got = _run(hip_ctx, kind, 1080, 1920)
    if kind == "subpixel" and not os.environ.get("ST_RECORD_FLOW_TIERS"):
        inner = got[100:-100, 100:-100]
        assert abs(np.median(inner[..., 0]) - 2.37) < 0.05 and abs(np.median(inner[..., 1]) + 1.61) < 0.05


def test_motion_batch_equals_single(hip_ctx):
    """All kinds in one call (pairs (2k, 2k+1) of a 14-frame buffer) equal the one-pair calls bit for bit."""
    frames = np.concatenate([np.stack(_case(k, 480, 640)[:2]) for k in MOTION_KINDS])
    d = torch.from_numpy(frames).cuda()
    pairs = [(2 * i, 2 * i + 1) for i in range(len(MOTION_KINDS))]
    got = hip_ctx.optical_flow(d, pairs=pairs)
    for i in range(len(MOTION_KINDS)):
        assert torch.equal(got[i], hip_ctx.optical_flow(d[2 * i:2 * i + 2])[0]), MOTION_KINDS[i]


# Recording (GPU box):
#   ST_RECORD_FLOW_TIERS=1 python -m pytest tests/test_flow_motion_gpu.py -q -m gpu -s -k "default or 1080p" | grep MOTION_TIER_RECORD
# and write {"<kind>_<h>x<w>": tier} into tests/golden/flow_motion_tiers.json.
