"""dd_iir_c64 / dd_iir_f64 launched many times on the same input: the output must be bit-identical from launch to launch (the block passes wait
for their LDS-DMA steps by COUNTED s_waitcnt vmcnt -- a step consumed before it has landed would show as a changing output)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from directdemod_amd import _hip, filters, comm
_hip.require_gpu()
rng = np.random.default_rng(5)
for n, reps in (((1 << 22) + 12345, 200), ((1 << 25) + 3, 60)):
    x = (rng.standard_normal(n, dtype=np.float32) + 1j * rng.standard_normal(n, dtype=np.float32)).astype(np.complex64)
    d = _hip.DevArray.from_host(x)
    w = comm._convert(d, np.complex128)
    for label, src in (("complex64 in place", d), ("complex128", w)):
        f = filters.butter(2048000, 20000.0, storeState=False)
        ref = f.applyOn(src).to_host()
        bad = 0
        t = time.perf_counter()
        for r in range(reps):
            y = f.applyOn(src)
            if r % 10 == 9 or r == reps - 1:
                bad += int(not np.array_equal(y.to_host(), ref))
            del y
        _hip.sync()
        print("n = %d, %s: %d launches, %d of %d compared outputs differ  (%.1f s)" % (n, label, reps, bad, (reps + 9) // 10, time.perf_counter() - t))
        assert bad == 0
print("iir soak ok")
