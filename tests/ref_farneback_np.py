"""Independent float64 re-derivation of the Farneback pipeline, used ONLY to cross-check the C oracle
(tests/test_oracle.py).  It is written against the published algorithm (G. Farneback, "Two-frame
motion estimation based on polynomial expansion", SCIA 2003) in the variant OpenCV implements --
Gaussian pyramid built from the full-resolution image, polynomial expansion by weighted least
squares with a separable Gaussian applicability, box-filtered normal equations, coarse-to-fine
refinement -- with library building blocks that share no code with oracle/oracle.c: scipy.ndimage
separable correlations, torch's bilinear resampling, an explicit 6x6 least-squares solve.
Everything is float64, so agreement with the float32 oracle is expected to ~1e-5 relative.
"""
import numpy as np
from scipy import ndimage


def gaussian_taps(ksize, sigma):
    if ksize == 3 and sigma <= 0:
        return np.array([0.25, 0.5, 0.25])
    s = sigma if sigma > 0 else ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
    x = np.arange(ksize) - (ksize - 1) * 0.5
    k = np.exp(-x * x / (2 * s * s))
    return k / k.sum()


def levels_for(h, w, num_levels=3, pyr_scale=0.5, min_size=32):
    k, scale = 0, 1.0
    while k < num_levels:
        scale *= pyr_scale
        if w * scale < min_size or h * scale < min_size:
            break
        k += 1
    return k


def resize_bilinear(img, dh, dw):
    """Half-pixel-centre bilinear resampling of (h,w[,c]) float64 (exact 2x decimation = 2x2 mean,
    as cv::resize does)."""
    import torch
    import torch.nn.functional as F
    a = img if img.ndim == 3 else img[..., None]
    h, w, _ = a.shape
    if (h, w) == (dh, dw):
        return img.copy()
    t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1)[None]
    if h == 2 * dh and w == 2 * dw:
        o = F.avg_pool2d(t, 2)
    else:
        o = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False)
    o = o[0].permute(1, 2, 0).numpy()
    return o if img.ndim == 3 else o[..., 0]


def pyramid_image(gray, k, pyr_scale=0.5):
    h, w = gray.shape
    scale = pyr_scale ** k
    sigma = (1.0 / scale - 1) * 0.5
    ksize = max(int(round(sigma * 5)) | 1, 3)
    taps = gaussian_taps(ksize, sigma)
    f = gray.astype(np.float64)
    f = ndimage.correlate1d(f, taps, axis=1, mode="mirror")      # BORDER_REFLECT_101
    f = ndimage.correlate1d(f, taps, axis=0, mode="mirror")
    return resize_bilinear(f, int(round(h * scale)), int(round(w * scale)))


def poly_expansion(img, n=5, sigma=1.2):
    """Weighted least-squares fit of f ~ c + b.x + x^T A x on every (2n+1)^2 neighbourhood with
    applicability g(x)g(y); returns (h,w,5) = [b_y, b_x, A_yy, A_xx, A_xy] in OpenCV's channel order
    (replicated borders)."""
    x = np.arange(-n, n + 1, dtype=np.float64)
    g = np.exp(-x * x / (2 * sigma * sigma))
    g /= g.sum()
    # basis (1, x, y, x^2, y^2, xy): moments of the image under the applicability
    def corr(ky, kx):
        t = ndimage.correlate1d(img, kx, axis=1, mode="nearest")
        return ndimage.correlate1d(t, ky, axis=0, mode="nearest")
    m = np.stack([corr(g, g), corr(g, g * x), corr(g * x, g), corr(g, g * x * x), corr(g * x * x, g), corr(g * x, g * x)], -1)
    X, Y = np.meshgrid(x, x)
    B = np.stack([np.ones_like(X), X, Y, X * X, Y * Y, X * Y], -1).reshape(-1, 6)
    Wt = (g[:, None] * g[None, :]).reshape(-1)
    G = B.T @ (Wt[:, None] * B)
    r = m @ np.linalg.inv(G).T                                    # (h,w,6): c, b_x, b_y, A_xx, A_yy, A_xy
    return np.stack([r[..., 2], r[..., 1], r[..., 4], r[..., 3], r[..., 5]], -1)


def update_matrices(R0, R1, flow):
    h, w, _ = R0.shape
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    fx, fy = xs + flow[..., 0], ys + flow[..., 1]
    x1, y1 = np.floor(fx).astype(int), np.floor(fy).astype(int)
    ax, ay = fx - x1, fy - y1
    inside = (x1 >= 0) & (x1 < w - 1) & (y1 >= 0) & (y1 < h - 1)
    xc, yc = np.clip(x1, 0, w - 2), np.clip(y1, 0, h - 2)
    w00, w01, w10, w11 = (1 - ax) * (1 - ay), ax * (1 - ay), (1 - ax) * ay, ax * ay
    Rw = (w00[..., None] * R1[yc, xc] + w01[..., None] * R1[yc, xc + 1] +
          w10[..., None] * R1[yc + 1, xc] + w11[..., None] * R1[yc + 1, xc + 1])
    r2 = np.where(inside, Rw[..., 0], 0.0)
    r3 = np.where(inside, Rw[..., 1], 0.0)
    r4 = np.where(inside, (R0[..., 2] + Rw[..., 2]) * 0.5, R0[..., 2])
    r5 = np.where(inside, (R0[..., 3] + Rw[..., 3]) * 0.5, R0[..., 3])
    r6 = np.where(inside, (R0[..., 4] + Rw[..., 4]) * 0.25, R0[..., 4] * 0.5)
    r2 = (R0[..., 0] - r2) * 0.5
    r3 = (R0[..., 1] - r3) * 0.5
    dx, dy = flow[..., 0], flow[..., 1]
    r2 = r2 + r4 * dy + r6 * dx
    r3 = r3 + r6 * dy + r5 * dx
    # confidence roll-off at the image border (5 pixels)
    border = np.array([0.14, 0.14, 0.4472, 0.4472, 0.4472])
    sx, sy = np.ones(w), np.ones(h)
    for i in range(min(5, w)):
        sx[i] *= border[i]
        sx[w - 1 - i] *= border[i]
    for i in range(min(5, h)):
        sy[i] *= border[i]
        sy[h - 1 - i] *= border[i]
    scale = sy[:, None] * sx[None, :]
    r2, r3, r4, r5, r6 = r2 * scale, r3 * scale, r4 * scale, r5 * scale, r6 * scale
    return np.stack([r4 * r4 + r6 * r6, (r4 + r5) * r6, r5 * r5 + r6 * r6, r4 * r2 + r6 * r3, r6 * r2 + r5 * r3], -1)


def box_solve(M, win=15):
    B = np.stack([ndimage.uniform_filter(M[..., c], size=win, mode="nearest") for c in range(5)], -1)
    g11, g12, g22, h1, h2 = [B[..., c] for c in range(5)]
    idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3)
    return np.stack([(g11 * h2 - g12 * h1) * idet, (g22 * h1 - g12 * h2) * idet], -1)


def farneback(prev_gray, next_gray, num_levels=3, pyr_scale=0.5, win=15, iters=3, poly_n=5, poly_sigma=1.2):
    h, w = prev_gray.shape
    flow = None
    for k in range(levels_for(h, w, num_levels, pyr_scale), -1, -1):
        I0, I1 = pyramid_image(prev_gray, k, pyr_scale), pyramid_image(next_gray, k, pyr_scale)
        lh, lw = I0.shape
        flow = np.zeros((lh, lw, 2)) if flow is None else resize_bilinear(flow, lh, lw) * (1.0 / pyr_scale)
        R0, R1 = poly_expansion(I0, poly_n, poly_sigma), poly_expansion(I1, poly_n, poly_sigma)
        for _ in range(iters):
            flow = box_solve(update_matrices(R0, R1, flow), win)
    return flow
