import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu_hardware():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import capi
    capi.build()
    capi.lib()
    return capi


@pytest.fixture(scope="session")
def gymnet():
    """The product package (directory gym.net_amd/), loaded through __graft_entry__'s loader."""
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def gpu_pkg(gymnet):
    """Product package on a box with a GPU.  Skips only when the machine has no GPU device node at
    all; on a GPU box a missing / unloadable HIP library is a hard failure, never a silent skip."""
    if not has_gpu_hardware():
        pytest.skip("no /dev/kfd: no AMD GPU on this machine")
    n = gymnet.device_count()
    assert n >= 1, "GPU box but HIP reports no device"
    return gymnet
