"""butter(2 048 000, 20 000) over 2^26 complex64 IQ samples (decode_funcube.py:160 shape): the class route (dd_iir_c64: samples read as they are)
against the widened route (complex128 copy + dd_iir_f64), alternating in one process; HIP-event-free wall clock around synchronised calls."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from directdemod_amd import _hip, filters, comm
_hip.require_gpu()
n = 1 << 26
rng = np.random.default_rng(1)
x = (rng.standard_normal(n, dtype=np.float32) + 1j * rng.standard_normal(n, dtype=np.float32)).astype(np.complex64)
d = _hip.DevArray.from_host(x)
f = filters.butter(2048000, 20000.0, storeState=False)
def run_c64():
    y = f.applyOn(d); _hip.sync(); del y
def run_wide():
    w = comm._convert(d, np.complex128); y = f.applyOn(w); _hip.sync(); del y, w
_hip.pool_trim(0)
for name, fn in (("c64 in place", run_c64), ("widened", run_wide)) * 3:
    fn()
    t = time.perf_counter()
    for _ in range(5):
        fn()
    print("  %-14s %.3f ms per pass" % (name, (time.perf_counter() - t) / 5 * 1e3))
