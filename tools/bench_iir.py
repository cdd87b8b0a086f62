#!/usr/bin/env python3
"""Side measurement: butter (6th order) over complex128 IQ on the device, block-parallel recurrence
(not the headline metric; quoted in DESIGN.md)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directdemod_amd import _hip, filters
_hip.require_gpu()
for log2n in (20, 24, 26):
    n = 1 << log2n
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex128)
    d = _hip.DevArray.from_host(x)
    f = filters.butter(2400000, 100000.0)
    y = f.applyOn(d); _hip.sync()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        y = f.applyOn(d)
    _hip.sync()
    dt = (time.perf_counter() - t0) / reps
    print("butter order 6, complex128, 2^%d samples: %.3f ms  %.1f MSamples/s  (%.1f GB/s of 32 B/sample)" % (log2n, dt * 1e3, n / dt / 1e6, 32 * n / dt / 1e9))
    del d, y
