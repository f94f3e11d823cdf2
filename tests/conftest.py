import os
import sys

import pytest

# the library's test hooks (UWIP_ACLAHE_TEST_FORCE_CL ...) are dead unless the process starts with this (csrc/uwip_internal.hpp)
os.environ["UWIP_TEST_HOOKS"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import _oracle
    return _oracle.load()


@pytest.fixture(scope="session")
def ctx():
    """One uwip context on cuda:0 for the whole GPU session (fails loudly without a device)."""
    import torch
    import uwimageproc_amd as uw
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    c = uw.Context(0)
    yield c
    c.close()
