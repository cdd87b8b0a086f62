#!/usr/bin/env python3
"""where the FFT kernel's output differs from the default kernel's: by row and lane of the block layout"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from directdemod_amd import _hip
import bench
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
n = 1 << 20
x = bench.make_input(torch, n, 0, dev, 3)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ntaps = 255
taps = np.ascontiguousarray(0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(ntaps) / (ntaps - 1)))
outs = {}
for kern in ("ab", "fft1k"):
    _hip.select_kernel(kern)
    out = torch.zeros(n, dtype=torch.float32, device=dev)
    h = C.c_void_p()
    fl = _hip.DD_CHAIN_FM | (0 if os.environ.get("NONCO") else _hip.DD_CHAIN_NCO)
    _hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), ntaps, _hip.cycles_q64(25000.0, 2400000), 1, fl), "create")
    got = C.c_int64(0)
    lib.dd_chain_reset(h, stream)
    _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, C.byref(got), stream), "process")
    torch.cuda.synchronize()
    outs[kern] = out.cpu().numpy()[:got.value]
    lib.dd_chain_destroy(h)
a, f = outs["ab"], outs["fft1k"]
d = np.abs((f - a + np.pi) % (2 * np.pi) - np.pi)
pa = 4064 - 1
nb = (len(d) - pa) // 3840 - 2
blk = d[pa:pa + nb * 3840].reshape(nb, 15, 256)
print("blocks", nb, "overall max", d.max(), "mean", d.mean())
print("by row :", np.array2string(blk.mean(axis=(0, 2)), precision=4))
print("by lane (first 72):", np.array2string(blk.mean(axis=(0, 1))[:72], precision=3))
print("by block (first 8):", np.array2string(blk.mean(axis=(1, 2))[:8], precision=4))
print("ab :", a[pa:pa + 8], a[pa + 256:pa + 260])
print("fft:", f[pa:pa + 8], f[pa + 256:pa + 260])
