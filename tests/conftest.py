import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def aof():
    """The product package (C ABI binding).  Builds libaof.so on first use."""
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ge.PKG_DIR, "csrc", "libaof.so")):
        ge.build()
    return ge.load_package()


@pytest.fixture(scope="session")
def synth(aof):
    import importlib
    return importlib.import_module("aero_optical_flow_amd.synth")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (checker)."""
    from oracle import pyoracle
    return pyoracle


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU")
    return torch.device("cuda:0")
