"""Generate golden vectors for DrawFlow by IMPORTING the reference implementation.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_draw_flow_golden.py
It loads /root/reference/scannertools/scannertools/vis.py through stub ``scannerpy`` and
``cv2`` modules (draw_flow itself only uses numpy), feeds it seeded frames and flow fields --
including negative flows, an all-zero field (0/0), a field whose maximum is negative and one
holding a NaN -- and stores inputs plus the reference's outputs in ``draw_flow_golden.npz``.
No reference source is copied.
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference/scannertools/scannertools/vis.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sp = types.ModuleType("scannerpy")
    sp.register_python_op = lambda **kw: (lambda f: f)
    sp.FrameType = object
    st = types.ModuleType("scannerpy.types")
    st.BboxList = object
    sp.types = st
    sys.modules["scannerpy"] = sp
    sys.modules["scannerpy.types"] = st
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("ref_vis", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    rng = np.random.default_rng(7)
    cases = {}
    h, w = 24, 40
    flows = {
        "mixed": (rng.standard_normal((h, w, 2)) * 3).astype(np.float32),
        "positive": np.abs(rng.standard_normal((h, w, 2)) * 5).astype(np.float32),
        "negative_max": (-np.abs(rng.standard_normal((h, w, 2))) - 0.25).astype(np.float32),
        "zeros": np.zeros((h, w, 2), np.float32),
        "large": (rng.standard_normal((h, w, 2)) * 1e4).astype(np.float32),
    }
    nanf = (rng.standard_normal((h, w, 2)) * 2).astype(np.float32)
    nanf[3, 5, 0] = np.nan
    flows["with_nan"] = nanf
    ragged = (rng.standard_normal((7, 13, 2)) * 2).astype(np.float32)
    flows["ragged_7x13"] = ragged
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, fl in flows.items():
            fh, fw = fl.shape[:2]
            frame = rng.integers(0, 256, (fh, fw, 3), dtype=np.uint8)
            out = ref.draw_flow(None, frame.copy(), fl.copy())
            assert out.dtype == np.uint8 and out.shape == (fh, 2 * fw, 3)
            cases[name + "_frame"] = frame
            cases[name + "_flow"] = fl
            cases[name + "_out"] = out
    np.savez_compressed(os.path.join(HERE, "draw_flow_golden.npz"), numpy_version=np.__version__, **cases)
    print("wrote", len(flows), "cases; numpy", np.__version__)


if __name__ == "__main__":
    main()
