#!/usr/bin/env python3
"""One-chunk passes of the C3 / C4 front ends (complex64) and of the C4 front end from raw u8, 2^26 samples: the quick A/B line for the
decimating kernels (tools/each_variant.sh python tools/bench_decim_one.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from directdemod_amd import _hip as hip
lib = hip.lib()
dev = torch.device("cuda:0")
n = 1 << 26
x = bench.make_input(torch, n, 0, dev, 1)
x8 = (x + 127.5).round().clamp(0, 255).to(torch.uint8).contiguous()
import scipy.signal
cases = (("C3 c64", scipy.signal.remez(127, [0, 100e3, 150e3, 4999999], [1, 0], fs=1e7), 50, 250000.0, 1e7, x, 0),
         ("C4 c64", scipy.signal.windows.blackmanharris(151), 34, 30000.0, 2048000.0, x, 0),
         ("C4 u8 ", scipy.signal.windows.blackmanharris(151), 34, 30000.0, 2048000.0, x8, hip.DD_CHAIN_U8_INPUT))
line = []
for name, taps, M, f, fs, src, fl in cases:
    taps = np.ascontiguousarray(taps, dtype=np.float64)
    h = C.c_void_p()
    hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), len(taps), hip.cycles_q64(f, fs), M, hip.DD_CHAIN_NCO | hip.DD_CHAIN_FM | fl))
    out = torch.empty(n // M + 8, dtype=torch.float32, device=dev)
    def run():
        hip.check(lib.dd_chain_reset(h, None))
        hip.check(lib.dd_chain_process(h, src.data_ptr(), out.data_ptr(), n, None, None))
    for _ in range(200):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(3):
        e0.record()
        for _ in range(500):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 500)
    line.append("%s %.4f (k%d)" % (name, min(ts), lib.dd_chain_last_kernel(h)))
    lib.dd_chain_destroy(h)
print("    " + "   ".join(line))
