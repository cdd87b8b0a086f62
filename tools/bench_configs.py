#!/usr/bin/env python3
"""Side measurements (not the headline bench): throughput of the fused chain on the
config-3 / config-4 front-end shapes and of the streaming ring feeder.  One line each."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from directdemod_amd import _hip, shard
    sys.path.insert(0, os.path.join(ROOT, "."))
    import bench
    _hip.require_gpu()
    dev = torch.device("cuda", 0)
    n = 1 << 26
    x = bench.make_input(torch, n, 0, dev, 7)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bh = 0.35875 - 0.48829 * np.cos(2 * np.pi * np.arange(151) / 150) + 0.14128 * np.cos(4 * np.pi * np.arange(151) / 150) \
        - 0.01168 * np.cos(6 * np.pi * np.arange(151) / 150)
    import scipy.signal as ss
    rz = ss.remez(127, [0, 100e3, 150e3, 4999999], [1, 0], fs=1e7)
    for name, taps, M, fs, f in (("C4 front end: BH151, /34, FM", bh, 34, 2048000, 30000.0),
                                ("C3 front end: remez127, /50, FM", rz, 50, 10000000, 250000.0),
                                ("BH151, /1, FM (MFMA path)", bh, 1, 2048000, 30000.0)):
        eng = shard.HipChainEngine(taps, f, fs, M, stream=stream)
        lib = _hip.lib()
        for _ in range(3):
            lib.dd_chain_reset(eng.h, stream)
            eng.process(x.data_ptr(), out.data_ptr(), n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        steps = 20
        e0.record()
        for _ in range(steps):
            lib.dd_chain_reset(eng.h, stream)
            eng.process(x.data_ptr(), out.data_ptr(), n)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        bps = 8.0 + 4.0 / M
        print(json.dumps({"config": name, "ms": round(ms, 4), "GS_per_s": round(n / ms / 1e6, 1),
                          "bytes_per_sample": round(bps, 3), "hbm_frac_of_8TBs": round(n * bps / (ms * 1e-3) / 8e12, 4)}))
        eng.close()
    # FIR only, complex64 output (16 B/sample), MFMA path
    outc = torch.empty((n, 2), dtype=torch.float32, device=dev)
    ham = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(255) / 254)
    eng = shard.HipChainEngine(ham, 25000.0, 2400000, 1, fm=False, stream=stream)
    lib = _hip.lib()
    for _ in range(20):
        lib.dd_chain_reset(eng.h, stream)
        eng.process(x.data_ptr(), outc.data_ptr(), n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        lib.dd_chain_reset(eng.h, stream)
        eng.process(x.data_ptr(), outc.data_ptr(), n)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(json.dumps({"config": "NCO + hamming255, /1, complex64 out (MFMA path)", "ms": round(ms, 4), "GS_per_s": round(n / ms / 1e6, 1),
                      "bytes_per_sample": 16.0, "hbm_frac_of_8TBs": round(n * 16.0 / (ms * 1e-3) / 8e12, 4)}))
    eng.close()
    del outc
    # streaming ring feeder: host-resident u8 -> decoded audio (PCIe inclusive)
    from directdemod_amd import source, stream as st
    nraw = 1 << 27
    raw = np.random.default_rng(1).integers(0, 256, size=(nraw, 2), dtype=np.uint8)
    src = source.IQarray(raw, 2048000)
    for M in (34,):
        st.stream_fm_chain(src, bh, 30000.0, M, chunk_size=20000000)
        t0 = time.perf_counter()
        o, r = st.stream_fm_chain(src, bh, 30000.0, M, chunk_size=20000000)
        dt = time.perf_counter() - t0
        print(json.dumps({"config": "streaming ring feeder, u8 over PCIe, BH151 /%d FM, 2^27 samples in 20 M chunks" % M,
                          "s": round(dt, 4), "GS_per_s": round(nraw / dt / 1e9, 2), "PCIe_GBps": round(2 * nraw / dt / 1e9, 1)}))


if __name__ == "__main__":
    main()
