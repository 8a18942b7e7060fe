import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build the native libraries when one is missing or the HIP library was made from other sources than this tree's
    (hipcc cross-compiles without a GPU).  Built artefacts are git-ignored AND withheld from the GPU box
    (.gpurunignore), so a GPU test session compiles what it tests on the machine it runs on;
    tests/test_host.py::test_library_matches_the_tree and test_hist_gpu.py::test_library_was_built_here check it."""
    import __graft_entry__ as g
    g.ensure_built()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """One line at the very end of the session's output: which native library the tests ran against -- the hash of the
    sources it was built from, the machine that compiled it and how long that took (the GPU box compiles what it tests)."""
    try:
        import json
        import socket
        from scannertools_amd import _native
        info = _native.build_info()
        rec = {}
        path = os.path.join(ROOT, "scannertools_amd", "lib", "build_record.json")
        if os.path.exists(path):
            rec = json.load(open(path))
        here = socket.gethostname()
        terminalreporter.write_line(
            "native build: libscannertools_hip.so src=%s (tree %s) compiled on %s at %s%s; this host: %s -> %s" % (
                info["src"], _native.source_hash(), info["host"], info["at"],
                " in %.1f s (%s, %s cpus)" % (rec["seconds"], rec["jobs"], rec["cpus"]) if rec.get("host") == info["host"] else "",
                here, "BUILT HERE" if info["host"] == here else "built elsewhere"))
    except Exception as e:  # never let the report line fail a session
        terminalreporter.write_line("native build: unknown (%r)" % (e,))


@pytest.fixture(scope="session")
def hip_ctx():
    """One HipContext for the GPU session.  Fails (not skips) if the native library is missing."""
    import torch
    from scannertools_amd.hip import HipContext
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    ctx = HipContext(0)
    yield ctx
    ctx.close()


# flow-iteration scheduling modes (the environment is read when a context is created):
#   default  kernel by launch size       march  marching kernel (k_flow_iter3) everywhere
#   tile     tile kernel everywhere
#   roles4 / roles5  role-split marching kernel (k_flow_iter_roles) with 4 / 5 column waves wherever a launch marches,
#                    and the role-split one-pass pyramid
FLOW_MODES = {"default": {}, "march": {"ST_ITER_TILE": "0", "ST_ITER_ROLES": "0", "ST_PYR_ROLES": "0"}, "tile": {"ST_ITER_TILE": "1"},
              "roles4": {"ST_ITER_TILE": "0", "ST_ITER_ROLES": "1", "ST_ROLES_NCW": "4", "ST_PYR_ROLES": "1"},
              "roles5": {"ST_ITER_TILE": "0", "ST_ITER_ROLES": "1", "ST_ROLES_NCW": "5", "ST_PYR_ROLES": "1"},
              # the luma conversion folded into the one-pass pyramid instead of the separate gray pass
              "foldgray": {"ST_PYR_FOLD_GRAY": "1"},
              # level-0 expansion straight from the gray frames, level 0 left out of the pyramid pass (calls above 16 pairs)
              "polyu8": {"ST_POLY_U8": "1"}, "polyf32": {"ST_POLY_U8": "0"},
              # kernels chosen as for a GPU shared with other kernel instances (no role-split workgroups for small calls)
              "concurrent": {"ST_CONCURRENT": "1"}}


def make_mode_ctx(mode, **kw):
    """A HipContext created under the environment of one scheduling mode."""
    from scannertools_amd.hip import HipContext
    env = FLOW_MODES[mode]
    keys = ("ST_ITER_TILE", "ST_PYR_FOLD_GRAY", "ST_ITER_ROLES", "ST_ROLES_NCW", "ST_ROLES_ROWS", "ST_PYR_ROLES", "ST_POLY_U8", "ST_CONCURRENT")
    saved = {k: os.environ.get(k) for k in keys}
    try:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        return HipContext(0, **kw)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(scope="session")
def mode_ctxs():
    """One context per scheduling mode of the flow iteration (every kernel instance reachable)."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    ctxs = {m: make_mode_ctx(m) for m in FLOW_MODES}
    yield ctxs
    for c in ctxs.values():
        c.close()


@pytest.fixture(params=list(FLOW_MODES))
def flow_ctx(request, mode_ctxs):
    return mode_ctxs[request.param]
