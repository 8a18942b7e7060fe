"""ShotBoundaries: host-side tail of the Histogram -> ShotBoundaries path.

Mirrors ``/root/reference/scannertools/scannertools/shot_detection.py:11-28``: same op name,
same arguments, same output contract (row 0 = list of boundary indices, rows 1.. = None), same
constants.  The reference op is a Python/numpy op that Scanner runs on the host with the whole
stream as one batch; so is this one.  The per-pair histogram distances are computed in one
vectorised pass instead of 3(n-1) scipy calls.  The windowed outlier test evaluates the
reference's ``np.mean`` / ``np.std`` over the same slices: full-width windows go through one
strided 2-D reduction (numpy reduces each row with the same pairwise summation as the 1-D call,
so means and standard deviations are bit-identical -- tests/test_host.py checks this against the
per-window loop and against the reference's golden outputs), the <= 999 edge windows one by one.
"""
from typing import Any, Sequence

import numpy as np

WINDOW_SIZE = 500          # shot_detection.py:7
BOUNDARY_BATCH = 10000000  # shot_detection.py:8

try:  # inside a real Scanner deployment the op registers itself exactly like the reference's
    import scannerpy
    from scannerpy.types import Histogram

    _register = scannerpy.register_python_op(name='ShotBoundaries', batch=BOUNDARY_BATCH)
except ImportError:  # standalone: plain function, driven by scannertools_amd.engine
    Histogram = Any

    def _register(fn):
        return fn


def histogram_diffs(histograms) -> np.ndarray:
    """diffs[0] = 0; diffs[i] = mean over the 3 channels of the Chebyshev (L-inf) distance
    between frame i-1's and frame i's channel histograms (shot_detection.py:14-18)."""
    h = np.asarray(histograms)
    n = len(h)
    if n == 0:
        return np.zeros(0)
    if h.ndim != 3 or h.shape[1] != 3:
        raise ValueError("histograms must be a sequence of 3 x bins arrays")
    h = h.astype(np.int64)
    d = np.abs(h[1:] - h[:-1]).max(axis=2)  # (n-1, 3) exact integers
    diffs = np.mean(d, axis=1) if n > 1 else np.zeros(0)
    return np.insert(diffs, 0, 0)


@_register
def shot_boundaries(config, histograms: Sequence[Histogram]) -> Sequence[Any]:
    diffs = histogram_diffs(histograms)
    n = len(diffs)

    # Do simple outlier detection to find boundaries between shots (shot_detection.py:21-26)
    boundaries = outlier_boundaries(diffs)

    return [boundaries] + [None for _ in range(len(histograms) - 1)]


def outlier_boundaries(diffs: np.ndarray) -> list:
    """i in [1, n) with diffs[i] - mean(win) > 2.5 * std(win), win = diffs[max(i-W,0):min(i+W,n)]."""
    diffs = np.ascontiguousarray(diffs, dtype=np.float64)
    n, W = len(diffs), WINDOW_SIZE
    flag = np.zeros(n, dtype=bool)
    if n >= 2 * W:
        # rows W..n-W have the full window diffs[i-W:i+W]: one strided (n-2W+1, 2W) view
        v = np.lib.stride_tricks.sliding_window_view(diffs, 2 * W)
        idx = np.arange(W, n - W + 1)
        flag[idx] = diffs[idx] - np.mean(v, axis=1) > 2.5 * np.std(v, axis=1)
        edge = list(range(1, W)) + list(range(n - W + 1, n))
    else:
        edge = range(1, n)
    for i in edge:
        window = diffs[max(i - W, 0):min(i + W, n)]
        flag[i] = diffs[i] - np.mean(window) > 2.5 * np.std(window)
    return [int(i) for i in np.nonzero(flag)[0]]



def shot_boundaries_device(ctx, histograms) -> Sequence[Any]:
    """The same op for histograms that are already on the GPU (a CUDA int32 (n, 3, bins) tensor, what HipContext.histogram
    returns): the distances and the windowed outlier test run on the device in numpy's summation order
    (st_shot_boundaries), so the boundary list equals `shot_boundaries`' bit for bit; same output contract.  An explicit
    choice of the caller (sc.ops.ShotBoundaries(..., device=DeviceType.GPU)): without a GPU it raises, it does not fall
    back to the host op."""
    n = int(histograms.shape[0])
    if n == 0:
        return [[]]   # what the host op returns for an empty stream ([boundaries] + [None] * -1)
    return [ctx.shot_boundaries(histograms, WINDOW_SIZE, 2.5)] + [None for _ in range(n - 1)]
