#!/usr/bin/env python3
"""Cycles per phase of k_chain_cos1k's row (library built with -DC1_TRACE: tools/mkvariant.sh N dd_cosfir -DC1_TRACE).
usage: LIB=build/variants/lib_N.so python tools/debug/cos_trace.py"""
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
x = bench.make_input(torch, n, 0, dev, 3)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
taps = np.ascontiguousarray(0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(255) / 254))
out = torch.zeros(n, dtype=torch.float32, device=dev)
h = C.c_void_p()
_hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), 255, _hip.cycles_q64(25000.0, 2400000), 1,
                               _hip.DD_CHAIN_NCO | _hip.DD_CHAIN_FM), "create")
got = C.c_int64(0)
for _ in range(200):
    lib.dd_chain_reset(h, stream)
    _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, C.byref(got), stream), "process")
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    lib.dd_chain_reset(h, stream)
    _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, C.byref(got), stream), "process")
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
NW, NPH = 2048, 8
buf = (C.c_ulonglong * (NW * (NPH + 2)))()
f = C.CDLL(_hip.LIB_PATH).dd_debug_cos1k_trace
f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
f.restype = C.c_int
_hip.check(f(buf, NW), "trace")
a = np.frombuffer(buf, dtype=np.uint64).reshape(NW, NPH + 2).astype(np.float64)
rows = a[:, NPH].sum()
names = ["wait loads, NCO, LDS writes", "prefetch, phase look-up, LDS reads", "pass A", "scan + window", "pass B", "discriminator",
         "LDS transposition + stores", "whole loop (incl. the unstored first row)"]
print("%s: %.4f ms per launch (with stamps); cycles per row and wave, %d rows" % (os.environ.get("LIB", "default"), ms, rows))
tot = a[:, :7].sum() / rows
for i, nm in enumerate(names):
    v = a[:, i].sum() / rows
    print("  %-44s %9.1f  %5.1f %%" % (nm, v, 100 * v / tot))
