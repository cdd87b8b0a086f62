"""k_chain_decim_w: where a chunked run differs from the one-call run (and both from a float64 direct form)  M K [fm] [u8]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from directdemod_amd import _hip as hip
M, K = int(sys.argv[1]), int(sys.argv[2])
fm_on = "fm" in sys.argv[3:]
u8 = "u8" in sys.argv[3:]
lib = hip.lib(); hip.require_gpu()
fs = 2048000
cuts = np.cumsum([0, 1, 2, K - 1, 3, 2047, 2048, 2049, 4096 + 5, 7, 30011, 1, 20000 + M])
L = int(cuts[-1])
rng = np.random.default_rng(5)
raw = rng.integers(0, 256, size=(L, 2), dtype=np.uint8)
x = ((raw[:, 0].astype(np.float32) - 127.5) + 1j * (raw[:, 1].astype(np.float32) - 127.5)).astype(np.complex64)
taps = np.ascontiguousarray(np.hamming(K) / np.sum(np.hamming(K))) if K > 2 else np.array([0.5, 0.5])
flags = hip.DD_CHAIN_NCO | (hip.DD_CHAIN_FM if fm_on else 0) | (hip.DD_CHAIN_U8_INPUT if u8 else 0)
src = hip.DevArray.from_host(raw.reshape(-1)) if u8 else hip.DevArray.from_host(x, dtype=np.complex64)
isz = 2 if u8 else 8
odt = np.float32 if fm_on else np.complex64
def run(bounds):
    h = C.c_void_p()
    hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), K, hip.cycles_q64(30000.0, fs), M, flags))
    outs = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        no = lib.dd_chain_out_count(h, int(b - a))
        o = hip.DevArray(max(1, no), odt)
        got = C.c_int64(0)
        hip.check(lib.dd_chain_process(h, src.ptr + isz * int(a), o.ptr, int(b - a), C.byref(got), None))
        outs.append(o.to_host()[:no])
    lib.dd_chain_destroy(h)
    return np.concatenate(outs)
got, one = run(cuts), run(np.array([0, L]))
bad = np.nonzero(got.view(np.uint32).reshape(len(got), -1) != one.view(np.uint32).reshape(len(one), -1))[0]
bad = np.unique(bad)
print("outputs", len(got), "differ at", len(bad), "first", bad[:20], "-> sample", (bad[:20] + (1 if fm_on else 0)) * M, "cuts", cuts)
if len(bad):
    i = bad[0]
    print("chunked", got[i - 1:i + 3], "one", one[i - 1:i + 3])
