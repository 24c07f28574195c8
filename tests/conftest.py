import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """Build (if stale) and load the product library; CPU-only machines can still load and inspect it."""
    from mosfhet_amd import build, engine
    build.build()
    return engine.lib()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
