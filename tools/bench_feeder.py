#!/usr/bin/env python3
"""Side measurement: the pinned-ring feeder (host-resident u8 -> decoded audio, PCIe inclusive) for
1, 2, 4, 8 staging-copy threads."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directdemod_amd import _hip, source, stream as st
_hip.require_gpu()
nraw = 1 << 28
raw = np.random.default_rng(1).integers(0, 256, size=(nraw, 2), dtype=np.uint8)
src = source.IQarray(raw, 2048000)
k = np.arange(151)
bh = 0.35875 - 0.48829 * np.cos(2 * np.pi * k / 150) + 0.14128 * np.cos(4 * np.pi * k / 150) - 0.01168 * np.cos(6 * np.pi * k / 150)
st.stream_fm_chain(src, bh, 30000.0, 34, chunk_size=20000000)
for th in (-1, 0, 1, 4):
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        o, r = st.stream_fm_chain(src, bh, 30000.0, 34, chunk_size=20000000, copy_threads=max(th, 1),
                                  staging="registered" if th < 0 else ("direct" if th == 0 else "pinned"))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(json.dumps({"config": "ring feeder, u8 over PCIe, BH151 /34 FM, 2^28 samples in 20 M chunks",
                      "staging": "registered in place (hipHostRegister windows)" if th < 0 else ("direct (pageable hipMemcpyAsync)" if th == 0 else "pinned slot"), "copy_threads": max(th, 0),
                      "s": round(best, 4), "GS_per_s": round(nraw / best / 1e9, 2), "host_to_device_GBps": round(2 * nraw / best / 1e9, 1)}))
