"""Shared synthetic-input helpers for the test-suite (deterministic, seeded)."""
import numpy as np


def random_frames(seed, n, h, w):
    return np.random.default_rng(seed).integers(0, 256, (n, h, w, 3), dtype=np.uint8)


def smooth_texture(seed, h, w, sigma=3.0):
    """Gaussian-filtered noise scaled to 0..255 (float64)."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    t = gaussian_filter(rng.standard_normal((h, w)), sigma)
    t = (t - t.min()) / (t.max() - t.min()) * 255.0
    return t


def translated_rgb_pair(seed, h, w, tx, ty, margin=24):
    """Two RGB frames cut from one texture so that next(x+tx, y+ty) == prev(x, y)."""
    tex = [smooth_texture(seed * 3 + c, h + 2 * margin, w + 2 * margin) for c in range(3)]
    f0 = np.stack([t[margin:margin + h, margin:margin + w] for t in tex], axis=-1)
    f1 = np.stack([t[margin - ty:margin - ty + h, margin - tx:margin - tx + w] for t in tex], axis=-1)
    return f0.astype(np.uint8), f1.astype(np.uint8)


def texture_stream(seed, n, h, w, margin=24, max_step=3):
    """n RGB frames: a texture under an integer random-walk translation (non-trivial flow)."""
    rng = np.random.default_rng(seed)
    tex = [smooth_texture(seed * 3 + c, h + 2 * margin, w + 2 * margin) for c in range(3)]
    pos = np.zeros(2, int)
    frames, steps = [], []
    for i in range(n):
        ox, oy = margin - pos[0], margin - pos[1]
        frames.append(np.stack([t[oy:oy + h, ox:ox + w] for t in tex], axis=-1).astype(np.uint8))
        step = rng.integers(-max_step, max_step + 1, 2)
        newpos = np.clip(pos + step, -margin, margin)
        steps.append(newpos - pos)
        pos = newpos
    return np.stack(frames), np.array(steps[:-1])


def planar5(a):
    """(h,w,5) -> (5,h,w) contiguous."""
    return np.ascontiguousarray(np.moveaxis(a, -1, 0))


def interleaved5(a):
    """(5,h,w) -> (h,w,5)."""
    return np.ascontiguousarray(np.moveaxis(a, 0, -1))
