import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip():
    """The HIP path, or a loud failure: GPU tests never fall back to anything else."""
    import torch
    from boxdreamer_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.load()
    return _lib
