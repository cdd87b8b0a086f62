#!/usr/bin/env python3
"""One-chunk passes of the decimating chain (BH151, NCO, FM, complex64 and raw u8, 2^26 samples) over a list of decimations, with the
kernel the library picks and with the tile kernels ("decimp"):  python tools/debug/decim_m_sweep.py 8 10 16 32 34 50 64"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch, scipy.signal
import bench
from directdemod_amd import _hip as hip
lib = hip.lib()
dev = torch.device("cuda:0")
n = 1 << 26
x = bench.make_input(torch, n, 0, dev, 1)
x8 = (x + 127.5).round().clamp(0, 255).to(torch.uint8).contiguous()
taps = np.ascontiguousarray(scipy.signal.windows.blackmanharris(151), dtype=np.float64)
for M in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 34, 64]:
    line = []
    for sel in (None, "decimp"):
        hip.select_kernel(sel)
        for src, fl in ((x, 0), (x8, hip.DD_CHAIN_U8_INPUT)):
            h = C.c_void_p()
            hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), len(taps), hip.cycles_q64(30000.0, 2048000.0), M, hip.DD_CHAIN_NCO | hip.DD_CHAIN_FM | fl))
            out = torch.empty(n // M + 8, dtype=torch.float32, device=dev)
            def run():
                hip.check(lib.dd_chain_reset(h, None))
                hip.check(lib.dd_chain_process(h, src.data_ptr(), out.data_ptr(), n, None, None))
            for _ in range(30):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                run()
            e1.record()
            torch.cuda.synchronize()
            line.append("%s k%d %.4f" % ("u8 " if fl else "c64", lib.dd_chain_last_kernel(h), e0.elapsed_time(e1) / 100))
            lib.dd_chain_destroy(h)
    hip.select_kernel(None)
    print("M = %2d   " % M + "   ".join(line))
