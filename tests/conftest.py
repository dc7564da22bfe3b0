import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _built_extension():
    """The shared library is a build artefact (git-ignored): compile it on first use so that a fresh checkout can run the
    suite without a separate build step (hipcc cross-compiles for gfx950 without a GPU, ~1-2 min)."""
    from funasr_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    yield


def _usable_cores() -> int:
    """min(affinity mask, cgroup quota): torch's default thread count follows the machine, and an OpenMP pool larger than
    the container's CPU quota runs the oracle ~100x slower."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


@pytest.fixture(scope="session", autouse=True)
def _torch_threads():
    import torch
    torch.set_num_threads(_usable_cores())
