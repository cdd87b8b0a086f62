import os
import sys
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# `-m gpu` runs the hot path FIRST (the chain kernels' parity against the oracle and the golden vectors, then the full-size
# properties, then determinism), the audio-rate stages after it, and the two tests that start bench.py as child processes
# (torch.distributed rendezvous, RCCL) LAST: whatever a box does to a rendezvous, the kernels' parity has been reported by then.
_ORDER = ("test_gpu_parity.py", "test_gpu_fullsize.py", "test_gpu_determinism.py", "test_gpu_audio.py",
          "test_gpu_bench_ranks.py", "test_gpu_bench_nccl.py")

# one test may not hold the session longer than this (pytest-timeout's 120 s ends a test whose Python code is waiting; a
# native call that never returns does not see the signal), and the session as a whole ends itself before a 1200 s outer limit
# would end it without a report.  Both leave through os._exit with a non-zero code after naming the test on stderr.
_HARD_TEST_S = float(os.environ.get("DD_TEST_HARD_LIMIT_S", "300"))
_HARD_SESSION_S = float(os.environ.get("DD_TEST_SESSION_LIMIT_S", "1080"))
_watch = {"t0": time.monotonic(), "test": None, "t_test": 0.0, "done": 0}


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = os.path.basename(str(item.fspath))
        return _ORDER.index(name) if name in _ORDER else -1        # CPU files keep their place in front
    items.sort(key=key)                                            # stable: the order inside a file stays


def _watchdog():
    while True:
        time.sleep(5.0)
        now = time.monotonic()
        cur = _watch["test"]
        if cur is not None and now - _watch["t_test"] > _HARD_TEST_S:
            sys.stderr.write("\n[conftest watchdog] %s has been running for %.0f s (limit %.0f s): ending the session, "
                             "%d tests had finished\n" % (cur, now - _watch["t_test"], _HARD_TEST_S, _watch["done"]))
            sys.stderr.flush()
            os._exit(70)
        if now - _watch["t0"] > _HARD_SESSION_S:
            sys.stderr.write("\n[conftest watchdog] session at %.0f s (limit %.0f s) in %s: ending it, %d tests had finished\n"
                             % (now - _watch["t0"], _HARD_SESSION_S, cur, _watch["done"]))
            sys.stderr.flush()
            os._exit(71)


def pytest_sessionstart(session):
    # the first `import torch` on a fresh box pages the image in (1-2 min): take it here, outside any test's time limit,
    # and say how long it and the first touch of the GPU took
    t0 = time.monotonic()
    try:
        import torch
    except Exception:                                              # CPU-only tests that do not need it still run
        torch = None
    t1 = time.monotonic()
    msg = "[conftest] import torch %.1f s" % (t1 - t0)
    markexpr = getattr(session.config.option, "markexpr", "") or ""
    if torch is not None and "not gpu" not in markexpr and torch.cuda.is_available():
        torch.zeros(1, device="cuda").item()
        msg += ", first GPU touch %.1f s" % (time.monotonic() - t1)
    sys.stderr.write(msg + "\n")
    _watch["t0"] = time.monotonic()
    threading.Thread(target=_watchdog, name="dd-test-watchdog", daemon=True).start()


def pytest_runtest_logstart(nodeid, location):
    _watch["test"], _watch["t_test"] = nodeid, time.monotonic()


def pytest_runtest_logfinish(nodeid, location):
    _watch["test"] = None
    _watch["done"] += 1


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    terminalreporter.write_line("[conftest] session wall time %.1f s, %d tests finished" %
                                (time.monotonic() - _watch["t0"], _watch["done"]))


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
