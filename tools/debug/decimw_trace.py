#!/usr/bin/env python3
"""Cycles per phase of an interior row of k_chain_decim_w (library built with -DDW_TRACE: tools/mkvariant.sh N dd_decimw -DDW_TRACE).
usage: [M=34] LIB=build/variants/lib_N.so python tools/debug/decimw_trace.py [u8]"""
import ctypes as C, os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
if os.environ.get("LIB"):
    os.environ["DD_LIB_PATH"] = os.environ["LIB"]
import torch
from directdemod_amd import _hip
import bench
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
n = 1 << 26
u8 = "u8" in sys.argv[1:]
M = int(os.environ.get("M", "34"))
x = bench.make_input(torch, n, 0, dev, 3)
if u8:
    x = (x + 127.5).round().clamp(0, 255).to(torch.uint8).contiguous()
k = np.arange(151)
taps = np.ascontiguousarray(0.35875 - 0.48829 * np.cos(2 * np.pi * k / 150) + 0.14128 * np.cos(4 * np.pi * k / 150) - 0.01168 * np.cos(6 * np.pi * k / 150))
out = torch.zeros(n // M + 8, dtype=torch.float32, device=dev)
h = C.c_void_p()
_hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), 151, _hip.cycles_q64(30000.0, 2048000), M,
                               _hip.DD_CHAIN_NCO | _hip.DD_CHAIN_FM | (_hip.DD_CHAIN_U8_INPUT if u8 else 0)), "create")
for _ in range(100):
    lib.dd_chain_reset(h, None)
    _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, None, None), "process")
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    lib.dd_chain_reset(h, None)
    _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, None, None), "process")
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
NW, NPH = 2048, 7
NTR = NPH + 6
buf = (C.c_ulonglong * (NW * NTR))()
f = C.CDLL(_hip.LIB_PATH).dd_debug_decimw_trace
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
f.restype = C.c_int
_hip.check(f(buf, NW), "trace")
a = np.frombuffer(buf, dtype=np.uint64).reshape(NW, NTR).astype(np.float64)
rows = a[:, NPH].sum()
names = ["wait for the loads, NCO, LDS writes", "issue the next row's loads", "tap loop (block-sum form: the sums up the lanes)", "discriminator, stores", "halo copy", "row geometry, row phasor",
         "block sums (block-sum form)"]
print("M = %d  " % M + "%s %s: %.4f ms per launch (with stamps); cycles per interior row and wave, %d rows; whole kernel %.0f cycles per wave" %
      (os.environ.get("LIB", "default"), "u8" if u8 else "c64", ms, rows, a[:, NPH + 1].mean()))
tot = a[:, :NPH].sum() / rows
for i, nm in enumerate(names):
    v = a[:, i].sum() / rows
    print("  %-40s %9.1f  %5.1f %%" % (nm, v, 100 * v / tot))
if a[:, NPH + 5].sum() > 0:
    runs = a[:, NPH + 5].mean()
    print("  per wave: %.1f rows in %.1f runs; kernel start -> first run %.0f cycles; per run: start -> the row before it staged %.0f, its block sums %.0f; rows %.0f of %.0f cycles" %
          (rows / NW, runs, a[:, NPH + 2].mean(), a[:, NPH + 3].sum() / a[:, NPH + 5].sum(), a[:, NPH + 4].sum() / a[:, NPH + 5].sum(), tot * rows / NW, a[:, NPH + 1].mean()))
