import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def fdr():
    """The product package (ctypes binding of libfdr.so).  Built in-tree if stale."""
    so = os.path.join(ROOT, PKG, "libfdr.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    return importlib.import_module(PKG)
