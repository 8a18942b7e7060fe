"""Pins the CPU oracle against REAL OpenCV output wherever `cv2` is importable ("first
action on any machine with OpenCV: dump golden vectors and pin the version").  Neither the
authoring container nor the GPU pool has OpenCV, so these tests skip there -- the oracle's parity
status stays "unpinned" for the OpenCV-backed ops until this file has run green somewhere; it is
written so that running it is all that is needed."""
import numpy as np
import pytest

import oracle
from util import random_frames, translated_rgb_pair

cv2 = pytest.importorskip("cv2")


def _gray_bits():
    """14-bit luma table up to OpenCV 3.4.2, 15-bit afterwards: detect from one pixel."""
    px = np.array([[[1, 2, 255]]], np.uint8)
    want = int(cv2.cvtColor(px, cv2.COLOR_BGR2GRAY)[0, 0])
    for bits in (15, 14):
        if int(oracle.cvt_color(px, oracle.COLOR_BGR2GRAY, gray_bits=bits)[0, 0, 0]) == want:
            return bits
    pytest.fail("neither luma table reproduces cv2.cvtColor (%s)" % cv2.__version__)


def test_histogram_and_color_against_opencv():
    f = random_frames(0, 1, 97, 131)[0]
    for bins in (16, 256):
        ref = np.stack([cv2.calcHist([f], [c], None, [bins], [0, 256]).ravel() for c in range(3)]).astype(np.int32)
        np.testing.assert_array_equal(oracle.hist_u8c3(f, bins), ref)
    bits = _gray_bits()
    np.testing.assert_array_equal(oracle.cvt_color(f, oracle.COLOR_BGR2GRAY, bits)[..., 0], cv2.cvtColor(f, cv2.COLOR_BGR2GRAY))
    np.testing.assert_array_equal(oracle.cvt_color(f, oracle.COLOR_RGB2GRAY, bits)[..., 0], cv2.cvtColor(f, cv2.COLOR_RGB2GRAY))
    for name in ("COLOR_BGR2RGB", "COLOR_BGR2HSV", "COLOR_BGR2YCrCb", "COLOR_RGB2YCrCb", "COLOR_YCrCb2BGR", "COLOR_YCrCb2RGB",
                 "COLOR_RGB2HSV", "COLOR_HSV2BGR", "COLOR_HSV2RGB", "COLOR_BGR2HSV_FULL", "COLOR_RGB2HSV_FULL",
                 "COLOR_HSV2BGR_FULL", "COLOR_HSV2RGB_FULL", "COLOR_BGR2YUV", "COLOR_RGB2YUV", "COLOR_YUV2BGR", "COLOR_YUV2RGB",
                 "COLOR_BGR2XYZ", "COLOR_RGB2XYZ", "COLOR_XYZ2BGR", "COLOR_XYZ2RGB",
                 "COLOR_BGR2HLS", "COLOR_RGB2HLS", "COLOR_HLS2BGR", "COLOR_HLS2RGB", "COLOR_BGR2HLS_FULL", "COLOR_RGB2HLS_FULL",
                 "COLOR_HLS2BGR_FULL", "COLOR_HLS2RGB_FULL"):
        np.testing.assert_array_equal(oracle.cvt_color(f, getattr(oracle, name)), cv2.cvtColor(f, getattr(cv2, name)), err_msg=name)
    # channel layout family (codes 0..3, 5, 9..31): every code on a source of the channel count it takes, the enum
    # value itself checked against cv2's
    from scannertools_amd._native import COLOR_CODES
    rng = np.random.default_rng(5)
    f4 = np.ascontiguousarray(np.concatenate([f, rng.integers(0, 256, f.shape[:2] + (1,), dtype=np.uint8)], axis=2))
    f4[:5, :5, 3] = 0
    f2 = rng.integers(0, 256, f.shape[:2] + (2,), dtype=np.uint8)
    srcs = {1: np.ascontiguousarray(f[..., :1]), 2: f2, 3: f, 4: f4}
    from util import cvt_source
    for name, code in COLOR_CODES.items():
        if 90 <= code <= 124:   # YUV 4:2:0 / 4:2:2 sources
            assert getattr(cv2, name) == code, name
            for hh, ww in ((6, 10), (54, 98)):
                src = cvt_source(rng, code, hh, ww)
                ref = cv2.cvtColor(src[..., 0] if src.shape[2] == 1 else src, code)
                got = oracle.cvt_color(src, code)
                np.testing.assert_array_equal(got[..., 0] if ref.ndim == 2 else got, ref, err_msg=name)
    for name, code in COLOR_CODES.items():
        if not (code <= 3 or code == 5 or 9 <= code <= 31):
            continue
        assert getattr(cv2, name) == code, name
        cin = next(c for c in (1, 2, 3, 4) if oracle.lib().orc_cvt_out_channels(code, c) > 0)
        ref = cv2.cvtColor(srcs[cin], code)
        got = oracle.cvt_color(srcs[cin], code, gray_bits=bits if code in (10, 11) else 15)
        np.testing.assert_array_equal(got.reshape(ref.shape) if got.shape[2] == 1 else got, ref, err_msg=name)


def test_resize_against_opencv():
    f = random_frames(1, 1, 97, 131)[0]
    modes = ((oracle.INTER_NEAREST, cv2.INTER_NEAREST), (oracle.INTER_LINEAR, cv2.INTER_LINEAR),
             (oracle.INTER_CUBIC, cv2.INTER_CUBIC), (oracle.INTER_AREA, cv2.INTER_AREA),
             (oracle.INTER_LANCZOS4, cv2.INTER_LANCZOS4))
    for (dw, dh) in ((426 // 4, 60), (65, 48), (262, 194), (131, 97), (43, 97), (200, 30)):
        for om, cm in modes:
            np.testing.assert_array_equal(oracle.resize_u8(f, dw, dh, om), cv2.resize(f, (dw, dh), interpolation=cm),
                                          err_msg="%dx%d mode %d" % (dw, dh, om))


def test_flow_consumers_against_opencv():
    rng = np.random.default_rng(2)
    fl = (rng.standard_normal((60, 80, 2)) * 9).astype(np.float32)
    mag, deg = cv2.cartToPolar(np.ascontiguousarray(fl[..., 0]), np.ascontiguousarray(fl[..., 1]), angleInDegrees=True)
    omag, odeg = oracle.cart_to_polar_deg(fl)
    # OpenCV builds with FMA contract the magnitude/polynomial differently from the scalar restatement
    assert np.abs(omag - mag).max() <= 1e-5 and np.abs(odeg - deg).max() <= 1e-3
    hm = cv2.calcHist([mag], [0], None, [64], [0, 64]).ravel().astype(np.int32)
    hd = cv2.calcHist([deg], [0], None, [64], [0, 360]).ravel().astype(np.int32)
    got = oracle.flow_hist(fl)
    assert np.abs(got[0] - hm).sum() <= 4 and np.abs(got[1] - hd).sum() <= 4      # vectors on a bin edge


def test_farneback_against_opencv():
    f0, f1 = translated_rgb_pair(9, 270, 480, 3, -2)
    bits = _gray_bits()
    g0, g1 = cv2.cvtColor(f0, cv2.COLOR_BGR2GRAY), cv2.cvtColor(f1, cv2.COLOR_BGR2GRAY)
    ref = cv2.calcOpticalFlowFarneback(g0, g1, None, 0.5, 3, 15, 3, 5, 1.2, 0)
    got = oracle.optical_flow_rgb(f0, f1, oracle.default_params(gray_bits=bits))
    rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert rel <= 1e-3 and np.abs(got - ref).max() <= 1e-2, (cv2.__version__, rel, np.abs(got - ref).max())
    # stage level: the pyramid's blur and the polynomial expansion are reachable through public API
    blur = cv2.GaussianBlur(g0.astype(np.float32), (9, 9), 1.5)
    assert np.abs(oracle.gaussian_blur(g0.astype(np.float32), 9, 1.5) - blur).max() <= 1e-3
