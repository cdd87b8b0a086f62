#!/usr/bin/env python3
"""A/B of the M = 1 FM chain kernels inside one process (switched through dd_debug_select_kernel: auto, ab, ws, fft1k, cos1k): outputs
compared against the default kernel, HIP-event time per launch.  KERNELS=ab,fft1k,cos1k NTAPS=255 N=26 U8=1 INPUT=A|B NORESET=1"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from directdemod_amd import _hip
import bench
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
n = 1 << int(os.environ.get("N", "26"))
if os.environ.get("INPUT", "B") == "A":       # SURVEY 8(d) input A: iid uniform u8 noise (source.py:117-118 on a dead channel), seed 1234
    x = (torch.from_numpy(np.random.default_rng(1234).integers(0, 256, size=(n, 2), dtype=np.uint8)).to(dev).float() - 127.5).contiguous()
else:
    x = bench.make_input(torch, n, 0, dev, 3)
U8 = bool(os.environ.get("U8"))
if U8:
    x = (x + 127.5).round().clamp(0, 255).to(torch.uint8).contiguous()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {v: k for k, v in vars(_hip).items() if k.startswith("DD_KERNEL_")}
kernels = os.environ.get("KERNELS", "ab,fft1k").split(",")
reps = int(os.environ.get("REPS", "200"))
for ntaps in [int(v) for v in os.environ.get("NTAPS", "255").split(",")]:
    taps = np.ascontiguousarray(0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(ntaps) / (ntaps - 1)))
    ref = None
    for rnd in range(int(os.environ.get("ROUNDS", "2"))):
        for kern in kernels:
            _hip.select_kernel(kern)
            out = torch.zeros(n, dtype=torch.float32, device=dev)
            h = C.c_void_p()
            _hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), ntaps, _hip.cycles_q64(25000.0, 2400000), 1,
                                           _hip.DD_CHAIN_NCO | _hip.DD_CHAIN_FM | (_hip.DD_CHAIN_U8_INPUT if U8 else 0)), "create")
            got = C.c_int64(0)
            def step():
                if not os.environ.get("NORESET"):          # NORESET=1: the stream continues (carried state: the output is not shifted by one)
                    lib.dd_chain_reset(h, stream)
                _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, C.byref(got), stream), "process")
            for _ in range(reps):
                step()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                step()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            msg = ""
            if ref is None:
                ref = out.clone()
            elif rnd == 0:
                d = (out[:got.value] - ref[:got.value] + np.pi).remainder(2 * np.pi) - np.pi
                ad = d.abs()
                msg = " vs %s: max |d| %.3g rad, mean %.3g, >1e-4: %d" % (kernels[0], ad.max().item(), ad.mean().item(), int((ad > 1e-4).sum().item()))
                if ad.max().item() > 1e-3:
                    bad = torch.nonzero(ad > 1e-3).flatten()
                    msg += " first bad %s" % bad[:8].tolist()
            print("%3d taps %s%-4s %-26s %.4f ms  %.1f GS/s  frac %.3f%s" % (ntaps, "u8 in, " if U8 else "", kern, names.get(lib.dd_chain_last_kernel(h)), ms, n / ms / 1e6,
                  (n * (6 if U8 else 12) / (ms * 1e-3)) / 8e12, msg), flush=True)
            lib.dd_chain_destroy(h)
