import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build the native libraries if a fresh checkout has none (hipcc cross-compiles without a GPU).
    Built artefacts are git-ignored; normally __graft_entry__.build() has produced them already."""
    lib = os.path.join(ROOT, "scannertools_amd", "lib")
    need = [os.path.join(lib, "libscannertools_hip.so"), os.path.join(lib, "libscannertools_imgproc.so"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "scannertools_amd", "csrc"), "-j4"], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "scannertools_amd", "scanner_kernels")], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def hip_ctx():
    """One HipContext for the GPU session.  Fails (not skips) if the native library is missing."""
    import torch
    from scannertools_amd.hip import HipContext
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    ctx = HipContext(0)
    yield ctx
    ctx.close()
