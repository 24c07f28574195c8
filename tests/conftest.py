import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "split_kernel: runs with the library's default kernel selection at N = 2048, l = 4 (two CUs per bootstrap for small batches: "
                                       "sums per accumulator component) and compares with the oracle in that order (tests/test_gpu_parity.py: reference_product_order)")


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
