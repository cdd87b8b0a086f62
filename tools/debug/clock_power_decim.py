#!/usr/bin/env python3
"""Shader clock and package power (rocm-smi, read-only) sampled while one decimating front end loops over 2^26 resident samples:
   CASE=C3|C4|C4u8 [LIB=build/variants/lib_N.so] [DUR=4] python tools/debug/clock_power_decim.py   (round 6: is k_chain_decim_b on the power cap?)"""
import ctypes as C, os, re, subprocess, sys, threading, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
if os.environ.get("LIB"):
    os.environ["DD_LIB_PATH"] = os.environ["LIB"]
import torch, scipy.signal
from directdemod_amd import _hip as hip
import bench
hip.require_gpu()
lib = hip.lib()
dev = torch.device("cuda", 0)
n = 1 << 26
case = os.environ.get("CASE", "C4")
x = bench.make_input(torch, n, 0, dev, 1)
fl = 0
if case == "C3":
    taps, M, f, fs = scipy.signal.remez(127, [0, 100e3, 150e3, 4999999], [1, 0], fs=1e7), 50, 250000.0, 1e7
else:
    taps, M, f, fs = scipy.signal.windows.blackmanharris(151), 34, 30000.0, 2048000.0
    if case == "C4u8":
        x = (x + 127.5).round().clamp(0, 255).to(torch.uint8).contiguous()
        fl = hip.DD_CHAIN_U8_INPUT
taps = np.ascontiguousarray(taps, dtype=np.float64)
h = C.c_void_p()
hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), len(taps), hip.cycles_q64(f, fs), M, hip.DD_CHAIN_NCO | hip.DD_CHAIN_FM | fl))
out = torch.empty(n // M + 8, dtype=torch.float32, device=dev)
samples, stop = [], False


def poll():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10)
            samples.append(r.stdout)
        except Exception as e:
            samples.append("error %r" % (e,))
        time.sleep(0.2)


def run(seconds):
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(300):
            lib.dd_chain_reset(h, None)
            hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, None, None))
        torch.cuda.synchronize()
        k += 300
    return k, time.perf_counter() - t0


run(1.0)
th = threading.Thread(target=poll)
th.start()
k, dt = run(float(os.environ.get("DUR", "4")))
stop = True
th.join()
sclk, pw = [], []
for s in samples[1:-1]:
    m = re.search(r"sclk[^,]*,?[^\d]*\((\d+)Mhz\)", s)
    for line in s.splitlines():
        cells = line.split(",")
        for c in cells:
            mm = re.match(r"\((\d+)Mhz\)", c.strip())
            if mm and "sclk" in s.splitlines()[0].split(",")[cells.index(c)].lower():
                sclk.append(int(mm.group(1)))
        if line.startswith("card"):
            hdr = s.splitlines()[0].split(",")
            for i, c in enumerate(cells):
                if i < len(hdr) and "power" in hdr[i].lower():
                    try:
                        pw.append(float(c))
                    except ValueError:
                        pass
print("%s %s (kernel id %d): %.4f ms per launch (wall, synchronised every 300); sclk %s MHz, package %s W over %d samples" %
      (os.environ.get("LIB", "product"), case, lib.dd_chain_last_kernel(h), dt / k * 1e3,
       ("%.0f" % np.median(sclk)) if sclk else "?", ("%.0f" % np.median(pw)) if pw else "?", len(samples)))
if not sclk or not pw:
    print(samples[len(samples) // 2])
