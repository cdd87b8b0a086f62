#!/usr/bin/env python3
"""M = 1 MFMA chain per tap class and output flavour (run once as is and once with DD_MFMA_KERNEL=ws for the A/B)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from directdemod_amd import _hip
import bench
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
n = 1 << 26
x = bench.make_input(torch, n, 0, dev, 3)
U8 = bool(os.environ.get("U8"))                             # raw interleaved uint8 I,Q input (2 B/sample) instead of complex64
if U8:
    x = (x + 127.5).round().clamp(0, 255).to(torch.uint8).contiguous()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
out = torch.empty(2 * n, dtype=torch.float32, device=dev)
names = {v: k for k, v in vars(_hip).items() if k.startswith("DD_KERNEL_")}
for ntaps in [int(v) for v in os.environ.get("NTAPS", "63,127,151,255").split(",")]:
    taps = np.ascontiguousarray(0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(ntaps) / (ntaps - 1)))
    for fm in (True, False):
        h = C.c_void_p()
        _hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), ntaps, _hip.cycles_q64(25000.0, 2400000), 1,
                                       _hip.DD_CHAIN_NCO | (_hip.DD_CHAIN_FM if fm else 0) | (_hip.DD_CHAIN_FORCE_DIRECT if os.environ.get("FORCE_DIRECT") else 0) | (_hip.DD_CHAIN_U8_INPUT if U8 else 0)), "create")
        got = C.c_int64(0)
        def step():
            lib.dd_chain_reset(h, stream)
            _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, C.byref(got), stream), "process")
        for _ in range(300):
            step()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            step()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 200
        print("%3d taps %s%-9s %-22s %.4f ms  %.1f GS/s" % (ntaps, "u8 in, " if U8 else "", "FM" if fm else "complex64", names.get(lib.dd_chain_last_kernel(h)), ms, n / ms / 1e6), flush=True)
        lib.dd_chain_destroy(h)
