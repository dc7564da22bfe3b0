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
