import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cpus() -> int:
    """CPUs this process may really use: min(affinity, cgroup cpu.max quota).  The GPU box shows 256 logical CPUs under a 16-CPU quota;
    torch's default of one thread per logical CPU makes every CPU-oracle forward of the suite crawl there (bench.py: usable_cpus)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    import torch
    torch.set_num_threads(_usable_cpus())


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
