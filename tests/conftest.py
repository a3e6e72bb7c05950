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


@pytest.fixture(autouse=True)
def _windowed_schedule_for_short_streams():
    """The tests' streams are short: make run_stream take the windowed schedule whenever it applies (>= 4 batches), so that
    it is the schedule under test; the per-batch schedule is covered by the exact-mode runs, the module API (update) and the
    tests that ask for it by name."""
    try:
        from tpnet_amd import RandomProjectionModule
    except Exception:
        yield
        return
    old = RandomProjectionModule.default_schedule
    RandomProjectionModule.default_schedule = "windowed"
    yield
    RandomProjectionModule.default_schedule = old


@pytest.fixture(autouse=True)
def _seeded_global_generators(request):
    """Every test starts from generators seeded by its own name: a module's P[0] and self.mlp are drawn from torch's global
    generator (models/TPNet.py:49-65), and a comparison that fails must fail again when it is re-run (the round-3 gradient
    tests drew different weights in every run).  Tests that seed for themselves are unaffected."""
    import zlib
    seed = zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF
    try:
        import numpy as np
        np.random.seed(seed)
        import torch
        torch.manual_seed(seed)
    except Exception:
        pass
    yield
