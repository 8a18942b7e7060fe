"""Generate golden vectors for ShotBoundaries by IMPORTING the reference implementation.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_shot_golden.py
It loads /root/reference/scannertools/scannertools/shot_detection.py through a 10-line
``scannerpy`` stub (the file only needs the ``register_python_op`` decorator and the
``scannerpy.types.Histogram`` annotation), feeds it synthetic histogram streams in the wire
format of scannertools/types.py:23-27 (3 x int32[bins] per frame) and stores inputs plus
the reference's row-0 output in ``shot_golden.npz``.  No reference source is copied.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/scannertools/scannertools/shot_detection.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sp = types.ModuleType("scannerpy")
    sp.register_python_op = lambda **kw: (lambda f: f)
    st = types.ModuleType("scannerpy.types")
    st.Histogram = object
    sp.types = st
    sys.modules["scannerpy"] = sp
    sys.modules["scannerpy.types"] = st
    spec = importlib.util.spec_from_file_location("ref_shot_detection", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def synth_stream(seed, n, bins, npix=640 * 480, mean_shot=250, jitter=0.02):
    """Per-shot Dirichlet colour distribution, per-frame multinomial noise + slow drift."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 3, bins), np.int32)
    i = 0
    cuts = []
    while i < n:
        length = int(rng.integers(max(2, mean_shot // 4), mean_shot * 2))
        base = rng.dirichlet(np.ones(bins) * 2.0, size=3)
        for t in range(i, min(n, i + length)):
            for c in range(3):
                p = base[c] * (1 + jitter * rng.standard_normal(bins))
                p = np.clip(p, 1e-9, None)
                p /= p.sum()
                out[t, c] = rng.multinomial(npix, p)
        i += length
        if i < n:
            cuts.append(i)
    return out, cuts


def main():
    ref = load_reference()
    cases = {}

    def add(name, h):
        frames = [[np.array(h[i, j]) for j in range(3)] for i in range(len(h))]
        res = ref.shot_boundaries(None, frames)
        assert len(res) == len(h) and all(r is None for r in res[1:])
        cases[name + "__hist"] = h.astype(np.int32)
        cases[name + "__bounds"] = np.array(res[0], np.int64)
        print(name, h.shape, "->", len(res[0]), "boundaries")

    for seed in (0, 1, 2):
        for n in ((2, 10, 999, 1000, 1001, 2000) if seed == 0 else (10, 1000)):
            add("s%d_n%d_b16" % (seed, n), synth_stream(seed, n, 16)[0])
    add("s0_n10000_b16", synth_stream(0, 10000, 16, npix=4096, mean_shot=1250, jitter=0.05)[0])
    add("s1_n600_b256", synth_stream(1, 600, 256, mean_shot=100)[0])
    # degenerate streams
    add("single_frame", synth_stream(3, 1, 16)[0])
    add("constant", np.repeat(synth_stream(4, 1, 16)[0], 50, axis=0))
    spike = np.repeat(synth_stream(5, 1, 16)[0], 400, axis=0)
    spike[200:] = synth_stream(6, 1, 16)[0][0]
    add("single_cut", spike)
    np.savez_compressed(os.path.join(HERE, "shot_golden.npz"), **cases)


if __name__ == "__main__":
    main()
