#!/usr/bin/env python3
"""k_chain_fft stage by stage against a NumPy model of the same data flow (dd_debug_fft_block)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from directdemod_amd import _hip
_hip.require_gpu()
lib = _hip.lib()
N = 4096
S1, S2 = 272, 289
def P(k): return 4 * (k & 3) + (k >> 2)
def bfly16(a, inv):
    n = np.arange(16)
    W = np.exp((2j if inv else -2j) * np.pi * np.outer(n, n) / 16)
    out = a @ W.T
    res = np.empty_like(a)
    for k in range(16): res[:, P(k)] = out[:, k]
    return res
def model(x, g, frac):
    K = len(g)
    t = np.arange(256); hi = t >> 4; lo = t & 15
    tw1 = np.exp(-2j * np.pi * np.outer(t, np.arange(16)) / 4096)
    tw2 = np.exp(-2j * np.pi * np.outer(lo, np.arange(16)) / 256)
    G = np.zeros(N, complex); G[:K] = g * np.exp(2j * np.pi * frac * np.arange(K))
    Hf = np.fft.fft(G)
    hp = np.empty((256, 16), complex)
    for k0 in range(16):
        for k1 in range(16):
            hp[16 * k0 + k1, :] = Hf[k0 + 16 * k1 + 256 * np.arange(16)] / N
    st = {}
    X1 = np.zeros(16 * S1, complex); X2 = np.zeros(16 * S2, complex)
    a = np.empty((256, 16), complex)
    for r in range(16): a[:, r] = x[t + 256 * r]
    st[0] = a.copy()
    a = bfly16(a, False); st[1] = a.copy()
    for k in range(1, 16): a[:, P(k)] *= tw1[:, k]
    st[2] = a.copy()
    for k in range(16): X1[t + S1 * k] = a[:, P(k)]
    for k in range(16): a[:, k] = X1[hi * S1 + lo + 16 * k]
    st[3] = a.copy()
    a = bfly16(a, False); st[4] = a.copy()
    for k in range(1, 16): a[:, P(k)] *= tw2[:, k]
    st[5] = a.copy()
    for k in range(16): X2[hi * S2 + lo + 17 * k] = a[:, P(k)]
    for k in range(16): a[:, k] = X2[hi * S2 + lo * 17 + k]
    st[6] = a.copy()
    a = bfly16(a, False); st[7] = a.copy()
    z = np.empty_like(a)
    for k in range(16): z[:, k] = a[:, P(k)] * hp[:, k]
    a = z; st[15] = a.copy()
    a = bfly16(a, True); st[8] = a.copy()
    for k in range(16): X2[hi * S2 + lo * 17 + k] = a[:, P(k)]
    for k in range(16): a[:, k] = X2[hi * S2 + lo + 17 * k]
    st[9] = a.copy()
    for k in range(1, 16): a[:, k] *= np.conj(tw2[:, k])
    st[10] = a.copy()
    a = bfly16(a, True); st[11] = a.copy()
    for k in range(16): X1[hi * S1 + lo + 16 * k] = a[:, P(k)]
    for k in range(16): a[:, k] = X1[t + S1 * k]
    st[12] = a.copy()
    for k in range(1, 16): a[:, k] *= np.conj(tw1[:, k])
    st[13] = a.copy()
    a = bfly16(a, True); st[14] = a.copy()
    return st
def run(x, g, fhz, fs, stage):
    xd = torch.from_numpy(np.ascontiguousarray(x.astype(np.complex64))).cuda()
    out = torch.zeros(256 * 16, dtype=torch.complex64, device="cuda")
    taps = np.ascontiguousarray(g, dtype=np.float64)
    _hip.check(lib.dd_debug_fft_block(xd.data_ptr(), taps.ctypes.data_as(C.POINTER(C.c_double)), len(taps), _hip.cycles_q64(fhz, fs), 1, stage,
                                      out.data_ptr(), None), "dbg")
    return out.cpu().numpy().reshape(256, 16)
if __name__ == "__main__":
    rng = np.random.default_rng(0)
    x = (rng.integers(0, 256, N) - 127.5) + 1j * (rng.integers(0, 256, N) - 127.5)
    g = np.hamming(255)
    st = model(x, g, 25000.0 / 2400000)
    order = [0, 1, 2, 3, 4, 5, 6, 7, 15, 8, 9, 10, 11, 12, 13, 14]
    for s in order:
        got = run(x, g, 25000.0, 2400000, s)
        ref = st[s]
        err = np.abs(got - ref)
        sc = np.abs(ref).max()
        bad = np.argwhere(err > 1e-4 * sc)
        print("stage %2d: max err %.3g (scale %.3g) rel %.2e  bad %d %s" % (s, err.max(), sc, err.max() / sc, len(bad),
              ("first (t,k): %s regs %s" % (bad[:4].tolist(), sorted(set(bad[:, 1].tolist())))) if len(bad) else ""))
