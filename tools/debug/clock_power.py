#!/usr/bin/env python3
"""Shader clock and package power (rocm-smi / amd-smi, read-only) sampled while one kernel runs in a loop:
   KERNEL=fft1k|ab|cos1k  [FLAVOUR=u8|cx]  [LIB=build/variants/lib_N.so]  python tools/debug/clock_power.py
   (FLAVOUR, round 6: the headline chain from raw u8 pairs / with complex64 output instead of angles)"""
import ctypes as C, os, subprocess, sys, threading, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
if os.environ.get("LIB"):                      # a variant build is LOADED in place of the product library, never copied over it
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
os.environ["DD_MFMA_KERNEL"] = os.environ.get("KERNEL", "fft1k")
flavour = os.environ.get("FLAVOUR", "")
flags = _hip.DD_CHAIN_NCO | (0 if flavour == "cx" else _hip.DD_CHAIN_FM)
if flavour == "u8":
    x = (x + 127.5).round().clamp(0, 255).to(torch.uint8).contiguous()
    flags |= _hip.DD_CHAIN_U8_INPUT
out = torch.zeros(2 * n if flavour == "cx" else n, dtype=torch.float32, device=dev)
h = C.c_void_p()
_hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), 255, _hip.cycles_q64(25000.0, 2400000), 1, flags), "create")
got = C.c_int64(0)
samples = []
stop = False


def poll():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10)
            samples.append((time.perf_counter(), r.stdout.strip().replace("\n", " | ")))
        except Exception as e:
            samples.append((time.perf_counter(), "error %r" % (e,)))
        time.sleep(0.2)


def run(seconds):
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(200):
            lib.dd_chain_reset(h, stream)
            _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, C.byref(got), stream), "process")
        torch.cuda.synchronize()
        k += 200
    return k, time.perf_counter() - t0


th = threading.Thread(target=poll)
th.start()
time.sleep(1.0)
t_idle = time.perf_counter()
k, dt = run(float(os.environ.get("DUR", "6")))
stop = True
th.join()
print("%s %s: %d launches in %.2f s = %.4f ms per launch (wall, synchronised every 200)" % (os.environ.get("LIB", "default"), os.environ["DD_MFMA_KERNEL"], k, dt, dt / k * 1e3))
for t, s in samples:
    print("  t=%+.2f s %s" % (t - t_idle, s[:400]))
