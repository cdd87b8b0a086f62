"""does FREEING a pageable host array that a large copy has just used stall the next device operation?  (the runtime pins such arrays in place;
free() of a large block returns its pages to the system.)  Per round: allocate a fresh numpy array, copy, free it, then time a tiny operation."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from directdemod_amd import _hip
from directdemod_amd._hip import lib, check
_hip.require_gpu(); _hip.wait_copy_warmup(); time.sleep(0.3)
dev = _hip.DevArray(64 << 20, np.uint8)
tiny = np.ones(4096, dtype=np.uint8)
def next_op():
    t = time.perf_counter()
    check(lib().dd_memcpy_h2d(dev.ptr, tiny.ctypes.data, 4096, None)); _hip.sync()
    return (time.perf_counter() - t) * 1e3
for mb in (0.5, 2, 8, 32):
    n = int(mb * (1 << 20))
    for kind in ("h2d", "d2h"):
        out = []
        for r in range(6):
            a = np.ones(n, dtype=np.uint8) if kind == "h2d" else np.empty(n, dtype=np.uint8)
            t = time.perf_counter()
            if kind == "h2d":
                check(lib().dd_memcpy_h2d(dev.ptr, a.ctypes.data, n, None))
            else:
                check(lib().dd_memcpy_d2h(a.ctypes.data, dev.ptr, n, None))
            _hip.sync()
            tc = (time.perf_counter() - t) * 1e3
            del a
            out.append("%.2f/%.2f" % (tc, next_op()))
        print("%5.1f MB %s: copy ms / next tiny op ms after the array is freed:  %s" % (mb, kind, "  ".join(out)))
keep = []
out = []
for r in range(6):
    a = np.ones(8 << 20, dtype=np.uint8); keep.append(a)
    check(lib().dd_memcpy_h2d(dev.ptr, a.ctypes.data, 8 << 20, None)); _hip.sync()
    out.append("%.2f" % next_op())
print("  8.0 MB h2d, arrays kept alive: next tiny op ms:  " + "  ".join(out))
