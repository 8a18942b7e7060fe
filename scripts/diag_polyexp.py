"""Polynomial expansion of the HIP library against the oracle at several sizes, stage entry point and whole-flow path
(diagnostic): python scripts/diag_polyexp.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle
from scannertools_amd.hip import HipContext, unpack_r
from util import translated_rgb_pair

with HipContext(0) as ctx:
    for (h, w) in [(48, 64), (240, 320), (203, 317), (480, 640), (480, 320), (240, 640), (540, 960), (1080, 1920)]:
        rng = np.random.default_rng(h * w)
        img = rng.random((h, w), dtype=np.float32) * 255
        got = ctx.polyexp(torch.from_numpy(img).cuda()).cpu().numpy()
        ref = oracle.polyexp(img)
        got = got.reshape(ref.shape) if got.shape != ref.shape else got
        bad = np.argwhere(~np.isclose(got, ref, rtol=0, atol=0))
        print("polyexp %4dx%-4d  equal %s  bad %d  first %s" % (h, w, np.array_equal(got, ref), len(bad), bad[:3].tolist()))
        f0, f1 = translated_rgb_pair(h, h, w, 3, -2)
        fl = ctx.optical_flow(torch.from_numpy(np.stack([f0, f1])).cuda()).cpu().numpy()[0]
        rf = oracle.optical_flow_rgb(f0, f1)
        print("   flow max abs diff %.3g  finite %s" % (np.abs(fl - rf).max(), np.isfinite(fl).all()))
