"""GPU parity of the sibling imgproc ops: Blur (HIP, through the C ABI and the
Scanner kernel class) vs the CPU oracle -- bit-exact (integer arithmetic spelled out in the
reference source, blur_kernel_cpu.cpp:62-79)."""
import numpy as np
import pytest
import torch

import oracle
from scannertools_amd.engine import CacheMode, Client, DeviceType, NamedVideoStream, PerfParams
from util import random_frames, texture_stream

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 8, 15, 31])
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
    for k in (3, 7):
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
