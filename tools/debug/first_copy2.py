"""what the first LARGE host-to-device copies of a process cost after the 4 KB warm-up copy (fresh process per MODE):
MODE=a: 246 MB allocation, then 40 MB pages;  MODE=b: a 1 MB, an 8 MB and a 32 MB copy from a scratch array first"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from directdemod_amd import _hip
from directdemod_amd._hip import lib, check
_hip.require_gpu()
_hip.wait_copy_warmup()
def T(label, fn):
    t = time.perf_counter(); r = fn(); _hip.sync(); print("  %-50s %9.2f ms" % (label, (time.perf_counter() - t) * 1e3)); return r
raw = np.zeros((122880000, 2), dtype=np.uint8); raw[::4096] = 1
mode = os.environ.get("MODE", "a")
if mode == "b":
    scr = np.ones(32 << 20, dtype=np.uint8)
    d = T("DevArray 32 MB", lambda: _hip.DevArray(32 << 20, np.uint8))
    for mb in (1, 8, 32):
        T("scratch h2d copy %d MB" % mb, lambda: check(lib().dd_memcpy_h2d(d.ptr, scr.ctypes.data, mb << 20, None)))
big = T("DevArray 245 MB", lambda: _hip.DevArray(122880000, _hip.IQ8))
for i in range(6):
    T("h2d copy 40 MB, page %d" % i, lambda: check(lib().dd_memcpy_h2d(big.ptr + 40000000 * i, raw.ctypes.data + 40000000 * i, 40000000, None)))
