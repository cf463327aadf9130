import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def engine_lib():
    from pogema_amd import _lib
    return _lib.load()


@pytest.fixture(autouse=True)
def _forget_walk_verdicts():
    """The zone walk's negative cache is per process; tests that assert what a walk finds must not inherit an earlier
    test's verdict."""
    yield
    mod = sys.modules.get("pogema_amd.buffers")
    if mod is not None:
        mod.WalkVerdicts.clear()
