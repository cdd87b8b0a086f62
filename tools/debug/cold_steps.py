#!/usr/bin/env python3
"""duration of each of the first launches of the C2 chain in a fresh process (HIP events per launch)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from directdemod_amd import _hip
import bench
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
n = 1 << 26
x = bench.make_input(torch, n, 0, dev, 3)
out = torch.zeros(n, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
time.sleep(float(os.environ.get("IDLE", "0")))
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
taps = np.ascontiguousarray(0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(255) / 254))
h = C.c_void_p()
_hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), 255, _hip.cycles_q64(25000.0, 2400000), 1, _hip.DD_CHAIN_NCO | _hip.DD_CHAIN_FM), "create")
got = C.c_int64(0)
N = 400
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
t0 = time.perf_counter()
ev[0].record()
for i in range(N):
    lib.dd_chain_reset(h, stream)
    _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, C.byref(got), stream), "process")
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
print("first 12 launches (ms):", " ".join("%.3f" % v for v in ms[:12]))
for a, b in ((12, 20), (20, 40), (40, 80), (80, 160), (160, 320), (320, 400)):
    print("launches %3d..%3d: mean %.4f ms" % (a, b, float(np.mean(ms[a:b]))))
print("wall for %d launches: %.1f ms" % (N, (time.perf_counter() - t0) * 1e3))
