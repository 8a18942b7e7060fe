"""GPU parity: Histogram op (HIP, through the C ABI) vs the CPU oracle -- bit-exact."""
import numpy as np
import pytest
import torch

import oracle
from util import random_frames

pytestmark = pytest.mark.gpu


def _oracle_batch(frames, bins):
    return np.stack([oracle.hist_u8c3(f, bins) for f in frames])


@pytest.mark.parametrize("h,w", [(1, 1), (5, 7), (37, 53), (480, 640), (1080, 1920)])
@pytest.mark.parametrize("bins", [16, 256])
def test_hist_random_matches_oracle(hip_ctx, h, w, bins):
    n = 3 if h * w > 100000 else 5
    frames = random_frames(h * 31 + w + bins, n, h, w)
    got = hip_ctx.histogram(torch.from_numpy(frames).cuda(), bins).cpu().numpy()
    assert got.dtype == np.int32 and got.shape == (n, 3, bins)
    np.testing.assert_array_equal(got, _oracle_batch(frames, bins))


@pytest.mark.parametrize("bins", [1, 2, 3, 10, 17, 100, 255])
def test_hist_odd_bin_counts(hip_ctx, bins):
    frames = random_frames(bins, 2, 61, 67)
    got = hip_ctx.histogram(torch.from_numpy(frames).cuda(), bins).cpu().numpy()
    np.testing.assert_array_equal(got, _oracle_batch(frames, bins))
    # definition check, independent of the oracle: bin = floor(v*bins/256)
    ref = np.stack([[np.bincount(((f[..., c].astype(np.int64) * bins) >> 8).ravel(), minlength=bins) for c in range(3)]
                    for f in frames])
    np.testing.assert_array_equal(got, ref)


def test_hist_known_answers(hip_ctx):
    h, w = 120, 200
    zero = np.zeros((1, h, w, 3), np.uint8)
    got = hip_ctx.histogram(torch.from_numpy(zero).cuda(), 16).cpu().numpy()
    assert (got[0, :, 0] == h * w).all() and got[0, :, 1:].sum() == 0
    # all-equal frame (worst-case LDS contention), distinct value per channel
    eq = np.empty((1, h, w, 3), np.uint8)
    eq[..., 0], eq[..., 1], eq[..., 2] = 7, 130, 255
    got = hip_ctx.histogram(torch.from_numpy(eq).cuda(), 256).cpu().numpy()
    for c, v in enumerate((7, 130, 255)):
        assert got[0, c, v] == h * w and got[0, c].sum() == h * w
    # ramp v = (x + y + c) % 256
    y, x = np.mgrid[0:h, 0:w]
    ramp = np.stack([(x + y + c) % 256 for c in range(3)], -1).astype(np.uint8)[None]
    got = hip_ctx.histogram(torch.from_numpy(ramp).cuda(), 256).cpu().numpy()
    ref = np.stack([np.bincount(ramp[0, ..., c].ravel(), minlength=256) for c in range(3)])
    np.testing.assert_array_equal(got[0], ref)


def test_hist_frame_list_unaligned(hip_ctx):
    """One buffer per Scanner element; buffers at odd byte offsets exercise the scalar head/tail."""
    h, w = 33, 47
    nb = 3 * h * w
    frames = random_frames(5, 4, h, w)
    big = torch.zeros(4 * (nb + 64) + 64, dtype=torch.uint8, device="cuda")
    views = []
    for i, off in enumerate((1, 7, 16, 35)):
        start = i * (nb + 64) + off
        v = big[start:start + nb].view(h, w, 3)
        v.copy_(torch.from_numpy(frames[i]).cuda())
        views.append(v)
    got = hip_ctx.histogram(views, 16).cpu().numpy()
    np.testing.assert_array_equal(got, _oracle_batch(frames, 16))


def test_hist_empty_batch(hip_ctx):
    out = hip_ctx.histogram(torch.zeros((0, 4, 4, 3), dtype=torch.uint8, device="cuda"), 16)
    assert tuple(out.shape) == (0, 3, 16)


def test_hist_full_size_properties(hip_ctx):
    """BASELINE config sizes: sum of bins == W*H per channel and 256 -> 16 fold identity."""
    n, h, w = 64, 1080, 1920
    g = torch.Generator(device="cuda").manual_seed(0)
    frames = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    h256 = hip_ctx.histogram(frames, 256)
    h16 = hip_ctx.histogram(frames, 16)
    assert (h256.sum(dim=2) == h * w).all()
    assert torch.equal(h256.view(n, 3, 16, 16).sum(dim=3).to(torch.int32), h16)
    # spot-check three frames against the oracle
    for i in (0, 31, 63):
        np.testing.assert_array_equal(h256[i].cpu().numpy(), oracle.hist_u8c3(frames[i].cpu().numpy(), 256))


def test_hist_rejects_bad_arguments(hip_ctx):
    from scannertools_amd.hip import StError
    f = torch.zeros((1, 4, 4, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(StError):
        hip_ctx.histogram(f, 0)
    with pytest.raises(StError):
        hip_ctx.histogram(f, 257)
    with pytest.raises(TypeError):
        hip_ctx.histogram(torch.zeros((1, 4, 4, 3), dtype=torch.uint8), 16)  # CPU tensor: no fallback


def test_shot_pipeline_on_device_stream(hip_ctx):
    """Config 3 in miniature: 1080p stream generated on the device with planted cuts ->
    Histogram (HIP) -> ShotBoundaries (host); compared with the planted cuts."""
    from scannertools_amd.shot_detection import shot_boundaries
    n, h, w = 240, 1080, 1920
    cuts = [60, 130, 131 + 40]
    g = torch.Generator(device="cuda").manual_seed(3)
    frames = torch.empty((n, h, w, 3), dtype=torch.uint8, device="cuda")
    base = None
    for i in range(n):
        if i == 0 or i in cuts:
            gain = 0.3 + 0.7 * torch.rand(3, device="cuda", generator=g)
            base = (torch.rand((h, w, 3), device="cuda", generator=g) * 255.0 * gain).to(torch.int16)
        frames[i] = (base + torch.randint(-3, 4, (h, w, 3), dtype=torch.int16, device="cuda", generator=g)).clamp_(0, 255).to(torch.uint8)
    hist = hip_ctx.histogram(frames, 16)
    assert (hist.sum(dim=2) == h * w).all()
    res = shot_boundaries(None, list(hist.cpu().numpy()))
    assert res[0] == cuts and all(r is None for r in res[1:])


_VARIANT_SCRIPT = r"""
import numpy as np, torch
from scannertools_amd.hip import HipContext
ctx = HipContext(0)
def ref(fr, bins):
    return torch.stack([torch.stack([torch.bincount((f[..., c].flatten().int() * bins) >> 8, minlength=bins) for c in range(3)]) for f in fr])
g = torch.Generator(device="cuda").manual_seed(3)
for (n, h, w) in ((5, 37, 53), (3, 480, 640), (2, 1080, 1920), (40, 1080, 1920)):
    fr = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    for bins in (256, 16, 100):
        assert torch.equal(ctx.histogram(fr, bins).long(), ref(fr, bins)), (n, h, w, bins)
# all-equal frames: every count of a channel lands on one counter (the packed kernel's 16-bit halves must not overflow):
# one 4K frame, and more 1080p frames than workgroup slots (the launcher would otherwise give a whole frame to a workgroup)
for (n, h, w) in ((1, 2160, 3840), (3, 2160, 3840), (600, 1080, 1920)):
    eq = torch.empty((n, h, w, 3), dtype=torch.uint8, device="cuda")
    eq[..., 0], eq[..., 1], eq[..., 2] = 255, 255, 0
    got = ctx.histogram(eq, 256)
    assert (got[:, 0, 255] == h * w).all() and (got[:, 1, 255] == h * w).all() and (got[:, 2, 0] == h * w).all(), (n, h, w)
    assert (got.sum(dim=2) == h * w).all()
    del eq
# unaligned frame list
big = torch.randint(0, 256, (4 * 5000,), dtype=torch.uint8, device="cuda", generator=g)
views = [big[i * 5000 + off: i * 5000 + off + 3 * 33 * 47].view(33, 47, 3) for i, off in enumerate((1, 7, 16, 35))]
assert torch.equal(ctx.histogram(views, 256).long(), ref(views, 256))
print("variant ok")
"""


@pytest.mark.parametrize("env", [{"ST_HIST_P2": "1"}, {"ST_HIST_VARIANT": "8"}, {"ST_HIST_VARIANT": "0"}, {"ST_HIST16": "0"}, {"ST_HIST_HALF": "1"}, {"ST_HIST_HALF": "0"}, {"ST_HIST_COMMIT": "atomic"}])
def test_hist_alternative_kernels(env):
    """The opt-in kernels (the switch is read once per process, so each runs in its own): the packed two-per-CU instance
    (ST_HIST_P2=1), eight copies per 256 threads, one copy per wave, and the general kernel at 16 bins -- against
    torch.bincount, with the all-equal frames that bound the packed kernel's half-counters."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, **env)
    e["PYTHONPATH"] = root + os.pathsep + e.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", _VARIANT_SCRIPT], env=e, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variant ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_library_was_built_here(hip_ctx):
    """The library this session loaded was compiled ON THIS MACHINE from this tree's sources: the .so files are withheld
    from the GPU box (.gpurunignore), so tests/conftest.py's session start ran the build (hipcc --offload-arch=gfx950 on
    the target); its record sits beside the libraries."""
    import json
    import os
    import socket
    from scannertools_amd import _native
    info = _native.build_info()
    assert info["src"] == _native.source_hash()
    assert info["host"] == socket.gethostname(), ("library compiled on %r, running on %r" % (info["host"], socket.gethostname()))
    rec = json.load(open(os.path.join(os.path.dirname(_native.LIB_PATH), "build_record.json")))
    assert rec["host"] == socket.gethostname() and rec["source_hash"] == info["src"]
    print("built on target: %s in %.1f s (%s, %s cpus)" % (info, rec["seconds"], rec["jobs"], rec["cpus"]))
