import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def select_kernel():
    """force one of the M = 1 chain kernels ("ab", "ws", "fft1k", None = by tap class) for the rest of the test through the
    library's debug entry dd_debug_select_kernel; the session's own choice (DD_MFMA_KERNEL at start-up) comes back afterwards"""
    from directdemod_amd import _hip
    before = os.environ.get("DD_MFMA_KERNEL")
    yield _hip.select_kernel
    _hip.select_kernel(before)
