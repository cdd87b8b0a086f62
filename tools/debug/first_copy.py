import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from directdemod_amd import _hip
from directdemod_amd._hip import lib, check
_hip.require_gpu()
def T(label, fn):
    t = time.perf_counter(); r = fn(); _hip.sync(); print("  %-50s %9.2f ms" % (label, (time.perf_counter() - t) * 1e3)); return r
raw = np.zeros((122880000, 2), dtype=np.uint8); raw[::4096] = 1
big = T("DevArray 245 MB", lambda: _hip.DevArray(122880000, _hip.IQ8))
small = np.ones(4096, dtype=np.uint8)
T("first h2d copy, 4 KB", lambda: check(lib().dd_memcpy_h2d(big.ptr, small.ctypes.data, 4096, None)))
T("h2d copy 40 MB (first big)", lambda: check(lib().dd_memcpy_h2d(big.ptr, raw.ctypes.data, 40000000, None)))
T("h2d copy 40 MB (second, other pages)", lambda: check(lib().dd_memcpy_h2d(big.ptr + 40000000, raw.ctypes.data + 40000000, 40000000, None)))
T("h2d copy 40 MB (same pages again)", lambda: check(lib().dd_memcpy_h2d(big.ptr, raw.ctypes.data, 40000000, None)))
T("h2d copy 165 MB (rest)", lambda: check(lib().dd_memcpy_h2d(big.ptr + 80000000, raw.ctypes.data + 80000000, 165760000, None)))
