"""pytest configuration: `gpu` marker, package loader, shared fixtures."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def thk():
    """The product package (ctypes face of libthk.so)."""
    graft.build_libthk()
    return graft.load_package()


@pytest.fixture(scope="session")
def ctx(thk):
    """A device context; only requested by gpu-marked tests."""
    c = thk.Context(0)
    yield c
    c.close()
