"""ms per pass of the C2 front end with complex64 output (offsetFreq + Hamming 255, no demod) on 2^26 device-resident samples,
and of the headline FM chain beside it:  python tools/bench_cx.py   (tools/each_variant.sh runs it per variant library)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
from directdemod_amd import _hip as hip

lib = hip.lib()
dev = torch.device("cuda:0")
n = 1 << 26
x = bench.make_input(torch, n, 0, dev, 1)
taps = np.ascontiguousarray(0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(255) / 254.0))
for name, flags, width in (("complex64 out", hip.DD_CHAIN_NCO, 2), ("FM out", hip.DD_CHAIN_NCO | hip.DD_CHAIN_FM, 1)):
    h = C.c_void_p()
    hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), 255, hip.cycles_q64(25000.0, 2.4e6), 1, flags))
    out = torch.empty((n, width), dtype=torch.float32, device=dev)
    def run():
        hip.check(lib.dd_chain_reset(h, None))
        hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, None, None))
    for _ in range(300):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for rep in range(3):
        e0.record()
        for _ in range(1000):
            run()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 1000)
    print(f"    {name}: kernel {lib.dd_chain_last_kernel(h)}  ms per pass {min(best):.4f} .. {max(best):.4f}")
    lib.dd_chain_destroy(h)
