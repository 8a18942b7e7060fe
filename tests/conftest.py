import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_ctx():
    """One HipContext for the GPU session.  Fails (not skips) if the native library is missing."""
    import torch
    from scannertools_amd.hip import HipContext
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    ctx = HipContext(0)
    yield ctx
    ctx.close()
