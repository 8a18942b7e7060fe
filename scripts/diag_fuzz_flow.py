"""Diagnose flow-fuzz outliers: GPU vs oracle vs the independent float64 derivation (tests/ref_farneback_np.py) at the
worst pixel of given fuzz cases.  python scripts/diag_fuzz_flow.py seed:h:w:a:b ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import oracle
import ref_farneback_np as R
from util import smooth_texture
from scannertools_amd.hip import HipContext

def cases(seed):
    rng = np.random.default_rng(77 + seed)
    for _ in range(4):
        h, w = int(rng.integers(2, 150)), int(rng.integers(2, 200))
        nf = int(rng.integers(2, 5))
        base = np.stack([smooth_texture(int(rng.integers(1 << 30)), h + 8, w + 8) for _ in range(3)], -1)
        frames = np.stack([base[dy:dy + h, dx:dx + w] for dy, dx in rng.integers(0, 9, (nf, 2))]).astype(np.uint8)
        pairs = [(int(a), int(b)) for a, b in rng.integers(0, nf, (int(rng.integers(1, 5)), 2))]
        yield h, w, frames, pairs

ctx = HipContext(0)
for spec in sys.argv[1:]:
    seed, H, W, A, B = (int(v) for v in spec.split(":"))
    for h, w, frames, pairs in cases(seed):
        if (h, w) != (H, W):
            continue
        got = ctx.optical_flow(torch.from_numpy(frames).cuda(), pairs=[(A, B)]).cpu().numpy()[0]
        ref = oracle.optical_flow_rgb(frames[A], frames[B])
        g0, g1 = oracle.gray_u8(frames[A]), oracle.gray_u8(frames[B])
        f64 = R.farneback(g0, g1)
        d = np.abs(got - ref).max(-1)
        y, x = np.unravel_index(np.argmax(d), d.shape)
        print("seed %d %dx%d pair (%d,%d): max|gpu-oracle| %.5f at (y=%d,x=%d); there gpu %s oracle %s f64 %s" % (
            seed, h, w, A, B, d.max(), y, x, got[y, x], ref[y, x], f64[y, x]))
        eg, eo = np.abs(got - f64).max(-1), np.abs(ref - f64).max(-1)
        print("   |gpu-f64| there %.5f  |oracle-f64| there %.5f ; whole field: max|gpu-f64| %.5f max|oracle-f64| %.5f ; pixels with |gpu-oracle|>5e-3: %d of %d" % (
            eg[y, x], eo[y, x], eg.max(), eo.max(), int((d > 5e-3).sum()), d.size))
        # conditioning at the worst pixel from the float64 pipeline's last iteration at level 0
        I0, I1 = R.pyramid_image(g0, 0), R.pyramid_image(g1, 0)
        R0, R1 = R.poly_expansion(I0), R.poly_expansion(I1)
        from scipy import ndimage
        M = R.update_matrices(R0, R1, f64)
        Bx = np.stack([ndimage.uniform_filter(M[..., c], size=15, mode="nearest") for c in range(5)], -1)
        g11, g12, g22, h1, h2 = [Bx[y, x, c] for c in range(5)]
        print("   level-0 box sums there: g11 %.4g g12 %.4g g22 %.4g h1 %.4g h2 %.4g det+1e-3 %.4g" % (g11, g12, g22, h1, h2, g11 * g22 - g12 * g12 + 1e-3))
