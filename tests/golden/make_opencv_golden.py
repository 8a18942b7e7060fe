#!/usr/bin/env python3
"""Run ONCE on any machine that has OpenCV (pip install opencv-python-headless numpy): writes
tests/golden/opencv_<version>.npz -- inputs and REAL OpenCV outputs for every OpenCV routine the oracle restates
(calcHist, cvtColor, resize in five modes, cartToPolar + calcHist, GaussianBlur, calcOpticalFlowFarneback at the sizes the
tests and the legacy pipeline use).  Commit the file: tests/test_oracle.py::test_oracle_against_opencv_golden then pins the
oracle against it on every machine (no OpenCV needed there), which is what turns the "parity unpinned" rows of DESIGN.md
section 2 into pinned ones.  Neither the authoring container nor the GPU pool has OpenCV, hence this script.

    python tests/golden/make_opencv_golden.py            # writes tests/golden/opencv_<cv2.__version__>.npz
    python -m pytest tests/test_against_opencv.py        # the same comparisons live, where cv2 is importable
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/ (util.py)


def main():
    import cv2
    from util import random_frames, translated_rgb_pair, texture_stream
    out = {"cv2_version": np.array(cv2.__version__)}
    f = random_frames(0, 1, 97, 131)[0]
    out["hist_in"] = f
    for bins in (16, 256):
        out["hist_%d" % bins] = np.stack([cv2.calcHist([f], [c], None, [bins], [0, 256]).ravel() for c in range(3)]).astype(np.int32)
    out["gray_bgr2gray"] = cv2.cvtColor(f, cv2.COLOR_BGR2GRAY)
    out["gray_rgb2gray"] = cv2.cvtColor(f, cv2.COLOR_RGB2GRAY)
    for name in ("COLOR_BGR2HSV", "COLOR_RGB2HSV", "COLOR_HSV2BGR", "COLOR_BGR2YCrCb", "COLOR_YCrCb2BGR", "COLOR_BGR2YUV",
                 "COLOR_YUV2BGR", "COLOR_BGR2XYZ", "COLOR_XYZ2BGR", "COLOR_BGR2HSV_FULL", "COLOR_HSV2BGR_FULL",
                 "COLOR_BGR2HLS", "COLOR_HLS2BGR", "COLOR_RGB2HLS_FULL", "COLOR_HLS2RGB_FULL"):
        out["cvt_" + name] = cv2.cvtColor(f, getattr(cv2, name))
    rs = random_frames(1, 1, 97, 131)[0]
    out["resize_in"] = rs
    for (dw, dh) in ((106, 60), (65, 48), (262, 194), (43, 97), (200, 30)):
        for mname in ("INTER_NEAREST", "INTER_LINEAR", "INTER_CUBIC", "INTER_AREA", "INTER_LANCZOS4"):
            out["resize_%dx%d_%s" % (dw, dh, mname)] = cv2.resize(rs, (dw, dh), interpolation=getattr(cv2, mname))
    big = random_frames(2, 1, 1080, 1920)[0]
    out["resize1080_in_seed"] = np.array(2)
    out["resize1080_426x240"] = cv2.resize(big, (426, 240), interpolation=cv2.INTER_LINEAR)
    rng = np.random.default_rng(2)
    fl = (rng.standard_normal((60, 80, 2)) * 9).astype(np.float32)
    mag, deg = cv2.cartToPolar(np.ascontiguousarray(fl[..., 0]), np.ascontiguousarray(fl[..., 1]), angleInDegrees=True)
    out["polar_in"], out["polar_mag"], out["polar_deg"] = fl, mag, deg
    out["flowhist_mag"] = cv2.calcHist([mag], [0], None, [64], [0, 64]).ravel().astype(np.int32)
    out["flowhist_deg"] = cv2.calcHist([deg], [0], None, [64], [0, 360]).ravel().astype(np.int32)
    # Farneback: the parameters of optical_flow_kernel_cpu.cpp:16 on cvtColor(BGR2GRAY) of RGB frames (the reference's quirk)
    cases = {"270x480": translated_rgb_pair(9, 270, 480, 3, -2), "240x426": translated_rgb_pair(4, 240, 426, -2, 1),
             "480x640": translated_rgb_pair(6, 480, 640, 4, 3), "97x131": translated_rgb_pair(3, 97, 131, 1, -1)}
    st, _ = texture_stream(5, 2, 203, 317)
    cases["203x317"] = (st[0], st[1])
    # motion that is not a whole-pixel shift: the pairs of tests/test_flow_motion_gpu.py (same seed), at its 480x640
    from util import MOTION_KINDS, motion_pair
    for kind in MOTION_KINDS:
        cases["%s_480x640" % kind] = motion_pair(kind, 3, 480, 640)
    for key, (f0, f1) in cases.items():
        g0, g1 = cv2.cvtColor(f0, cv2.COLOR_BGR2GRAY), cv2.cvtColor(f1, cv2.COLOR_BGR2GRAY)
        out["fb_%s_f0" % key], out["fb_%s_f1" % key] = f0, f1
        out["fb_%s_flow" % key] = cv2.calcOpticalFlowFarneback(g0, g1, None, 0.5, 3, 15, 3, 5, 1.2, 0)
        out["fb_%s_blur9" % key] = cv2.GaussianBlur(g0.astype(np.float32), (9, 9), 1.5)
    path = os.path.join(HERE, "opencv_%s.npz" % cv2.__version__)
    np.savez_compressed(path, **out)
    print("wrote", path, "(%d arrays)" % len(out))


if __name__ == "__main__":
    main()
