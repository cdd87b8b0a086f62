#!/usr/bin/env python3
"""Side measurement: fused u8 ingest (device-resident raw I,Q bytes, 2 B/sample) through the decimating chain."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from directdemod_amd import _hip, shard
_hip.require_gpu()
dev = torch.device("cuda", 0)
n = 1 << 26
raw = torch.randint(0, 256, (n, 2), dtype=torch.uint8, device=dev)
out = torch.empty(n, dtype=torch.float32, device=dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
k = np.arange(151)
bh = 0.35875 - 0.48829 * np.cos(2 * np.pi * k / 150) + 0.14128 * np.cos(4 * np.pi * k / 150) - 0.01168 * np.cos(6 * np.pi * k / 150)
for M in (34, 1):
    eng = shard.HipChainEngine(bh, 30000.0, 2048000, M, u8=True, stream=stream)
    lib = _hip.lib()
    for _ in range(5):
        lib.dd_chain_reset(eng.h, stream)
        eng.process(raw.data_ptr(), out.data_ptr(), n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.dd_chain_reset(eng.h, stream)
        eng.process(raw.data_ptr(), out.data_ptr(), n)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(json.dumps({"config": "u8 device-resident, BH151, /%d, FM" % M, "ms": round(ms, 4), "GS_per_s": round(n / ms / 1e6, 1)}))
    eng.close()
