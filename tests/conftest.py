import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    """One device context for the whole GPU session (fails, not skips, when the library is missing)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device in this environment")
    from midoridb_amd.dev import DeviceCtx
    ctx = DeviceCtx(0)
    yield ctx
    ctx.close()
