import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def ctx():
    # torch first: its bundled HIP runtime and the one libcubeslam_hip.so links must be the same copy in one process; when the
    # library's copy is loaded first torch.cuda reports no usable device (bench.py imports torch first for the same reason)
    import torch
    torch.cuda.is_available()
    from cube_slam_amd import _lib
    c = _lib.Context(0)  # raises if the HIP library / device is missing: GPU tests must not silently fall back
    yield c
    c.close()
