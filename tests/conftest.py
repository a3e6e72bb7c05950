import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def hip_lib():
    """Builds (if needed) and loads the C-ABI library."""
    so = os.path.join(ROOT, "tpnet_amd", "libtpnet_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    from tpnet_amd import _lib
    return _lib.load()
