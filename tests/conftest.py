import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The built C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    from qed_splatter_amd.build import build_lib
    from qed_splatter_amd import _lib
    build_lib()
    return _lib.load()


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from qed_splatter_amd.build import LIB_PATH
    assert LIB_PATH.exists(), "libqed_splat.so must be built before the GPU tests (python __graft_entry__.py)"
    return torch.device("cuda:0")
