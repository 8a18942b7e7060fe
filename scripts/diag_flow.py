"""Diagnostics (not a test): HIP flow vs oracle error statistics on a few inputs."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle
from scannertools_amd.hip import HipContext
from util import translated_rgb_pair

ctx = HipContext(0)
def cu(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
def stats(name, got, ref):
    d = np.abs(got - ref)
    i = np.unravel_index(d.argmax(), d.shape)
    print("%-28s relL2 %.3e  maxabs %.3e at %s (ref %s got %s)  mean|d| %.3e  max|ref| %.2f" % (
        name, np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30), d.max(), i, ref[i], got[i], d.mean(), np.abs(ref).max()))
for (h, w, tx, ty, seed) in [(240, 320, 0, 0, 5), (240, 320, 3, -2, 240), (480, 640, 3, -2, 480), (1080, 1920, 4, 3, 21)]:
    f0, f1 = translated_rgb_pair(seed, h, w, tx, ty)
    t = time.time(); ref = oracle.optical_flow_rgb(f0, f1); tc = time.time() - t
    got = ctx.optical_flow(cu(np.stack([f0, f1]))).cpu().numpy()[0]
    stats("%dx%d t=(%d,%d)" % (w, h, tx, ty), got, ref)
    d = np.abs(got - ref).max(axis=2)
    print("   px with |d|>1e-3: %d; >1e-4: %d of %d; oracle %.2fs" % ((d > 1e-3).sum(), (d > 1e-4).sum(), d.size, tc))
