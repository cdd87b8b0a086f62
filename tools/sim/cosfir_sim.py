#!/usr/bin/env python3
"""float32 model of the cosine-series running-sum form of the M = 1 chain (the arithmetic of tools/ubench/cosfir_arith.hip with the
fresh-window scan of DESIGN.md 4.2d), against the float64 definition: NCO (comm.py:63-78) -> Hamming 255 FIR with a ones history
(filters.py:45,199) -> FM (demod_fm.py:40-49).  Runs on the CPU; prints the FIR error relative to max|y| and the FM angle error
statistics (the H3 mask of the tests)."""
import sys
import numpy as np

f32, c64 = np.float32, np.complex64
SCAN_DT = np.complex64      # np.complex128: the scan over lane totals and the window differences in float64 (6 values per row and lane)
SCAN_F = np.float32
K = 255
FS, F0 = 2400000.0, 25000.0


def input_B(n, seed=3):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / FS
    ph = 2 * np.pi * 25e3 * t + 5.0 * np.sin(2 * np.pi * 1e3 * t)
    re = 60 * np.cos(ph) + 4 * rng.standard_normal(n)
    im = 60 * np.sin(ph) + 4 * rng.standard_normal(n)
    return (np.clip(np.round(re + 127.5), 0, 255) - 127.5) + 1j * (np.clip(np.round(im + 127.5), 0, 255) - 127.5)


def input_A(n, seed=4):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, n) - 127.5) + 1j * (rng.integers(0, 256, n) - 127.5)


def oracle(x):
    n = len(x)
    xt = x.astype(c64).astype(np.complex128) * np.exp(-2j * np.pi * F0 * np.arange(n) / FS)
    xt = xt.astype(c64).astype(np.complex128)                 # the reference rounds the product to complex64 in place (comm.py:77)
    w = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(K) / (K - 1))
    full = np.concatenate([np.ones(K - 1, dtype=np.complex128), xt])
    y = np.convolve(full, w)[K - 1:K - 1 + n]
    return y, np.angle(y[1:] * np.conj(y[:-1]))


def rot(c, s, C, S):
    """(C, S) <- A (C, S) in the operands' precision (fma chains are modelled as separately rounded multiply-adds: pessimistic)"""
    dt = np.result_type(C, S)
    return (c * C - s * S).astype(dt), (s * C + c * S).astype(dt)


def model(x, horizon_rows=True):
    n = len(x)
    rows = n // 1024
    assert rows * 1024 == n
    phi = 2 * np.pi / (K - 1)
    c, s = f32(np.cos(phi)), f32(np.sin(phi))
    # NCO: exact phase per sample, product rounded once (the kernel's phasor is a product of three table factors: ~1e-7)
    xt = (x.astype(c64) * np.exp(-2j * np.pi * ((F0 / FS * np.arange(n)) % 1.0)).astype(c64)).astype(c64)
    hist = np.ones(1024, dtype=c64)                            # row -1: the ones history (only its last 254 samples matter)
    hist[:1024 - 254] = 0
    X = np.concatenate([hist, xt]).reshape(rows + 1, 64, 16)    # [row][lane][i]
    # pass A: lane totals T (C, S, R) in the end-of-lane frame, un-combed
    C = X[:, :, 0].copy(); S = np.zeros_like(C); R = X[:, :, 0].copy()
    for i in range(1, 16):
        C, S = rot(c, s, C, S)
        C = (C + X[:, :, i]).astype(c64)
        R = (R + X[:, :, i]).astype(c64)
    # inclusive weighted scan over the 64 lanes of a row (Kogge-Stone, float32)
    P_C, P_S, P_R = C.astype(SCAN_DT), S.astype(SCAN_DT), R.astype(SCAN_DT)
    for sh in (1, 2, 4, 8, 16, 32):
        wc, ws = SCAN_F(np.cos(16 * phi * sh)), SCAN_F(np.sin(16 * phi * sh))
        sc = np.zeros_like(P_C); ss = np.zeros_like(P_S); sr = np.zeros_like(P_R)
        sc[:, sh:] = P_C[:, :-sh]; ss[:, sh:] = P_S[:, :-sh]; sr[:, sh:] = P_R[:, :-sh]
        rc, rs = rot(wc, ws, sc, ss)
        P_C = (P_C + rc).astype(SCAN_DT); P_S = (P_S + rs).astype(SCAN_DT); P_R = (P_R + sr).astype(SCAN_DT)
    # windowed state at the END of lane L:  V_L = P_L - A^256 P_{L-16} - A^255 (xt[16 (L - 15)], 0), previous row for L < 16
    w256c, w256s = SCAN_F(np.cos(256 * phi)), SCAN_F(np.sin(256 * phi))
    w255c, w255s = SCAN_F(np.cos(255 * phi)), SCAN_F(np.sin(255 * phi))
    V_C = np.zeros((rows + 1, 64), SCAN_DT); V_S = np.zeros_like(V_C); V_R = np.zeros_like(V_C)
    Z = np.zeros(64, SCAN_DT)
    for r in range(0, rows + 1):
        pC, pS, pR = (P_C[r - 1], P_S[r - 1], P_R[r - 1]) if r else (Z, Z, Z)
        for L in range(64):
            if L >= 16:
                pc, ps, pr = P_C[r, L], P_S[r, L], P_R[r, L]
                qc, qs, qr = P_C[r, L - 16], P_S[r, L - 16], P_R[r, L - 16]
            else:
                # continue the previous row's prefix: P_L + A^{16 (L + 1)} (P_63' - A^{16 (15 - L)} P_{48+L}')
                a = 16 * (15 - L)
                tc, ts = rot(SCAN_F(np.cos(a * phi)), SCAN_F(np.sin(a * phi)), pC[48 + L], pS[48 + L])
                tc = SCAN_DT(pC[63] - tc); ts = SCAN_DT(pS[63] - ts); tr = SCAN_DT(pR[63] - pR[48 + L])
                a = 16 * (L + 1)
                uc, us = rot(SCAN_F(np.cos(a * phi)), SCAN_F(np.sin(a * phi)), tc, ts)
                pc, ps, pr = SCAN_DT(P_C[r, L] + uc), SCAN_DT(P_S[r, L] + us), SCAN_DT(P_R[r, L] + tr)
                qc = qs = qr = SCAN_DT(0)
            e = X[r, L - 15, 0] if L >= 15 else (X[r - 1, 64 + L - 15, 0] if r else SCAN_DT(0))
            rc, rs = rot(w256c, w256s, qc, qs)
            V_C[r, L] = SCAN_DT(SCAN_DT(pc - rc) - SCAN_DT(w255c * e)); V_S[r, L] = SCAN_DT(SCAN_DT(ps - rs) - SCAN_DT(w255s * e)); V_R[r, L] = SCAN_DT(SCAN_DT(pr - qr) - e)
    # pass B: from the state at the start of the lane (V of the lane before), comb steps
    flatV = lambda V: np.concatenate([V[:-1, 63:64], V[1:, :63]], axis=1)          # state entering lane L of rows 1..: V_{L-1}
    C, S, R = flatV(V_C).astype(c64), flatV(V_S).astype(c64), flatV(V_R).astype(c64)
    Xc = X[1:]
    D = np.concatenate([X[:-1].reshape(rows, 1024)[:, 1024 - 255:], X[1:].reshape(rows, 1024)[:, :1024 - 255]], axis=1).reshape(rows, 64, 16)
    y = np.zeros((rows, 64, 16), c64)
    a0, a1 = f32(0.54), f32(-0.46)
    for i in range(16):
        C = (C - D[:, :, i]).astype(c64)
        C, S = rot(c, s, C, S)
        C = (C + Xc[:, :, i]).astype(c64)
        R = (R + c64(Xc[:, :, i] - D[:, :, i])).astype(c64)
        y[:, :, i] = (a0 * R + a1 * C).astype(c64)
    y = y.reshape(-1)
    ang = np.angle(y[1:].astype(np.complex128) * np.conj(y[:-1].astype(np.complex128)))
    return y, ang


def report(name, x):
    yo, ao = oracle(x)
    ym, am = model(x)
    e = np.abs(ym - yo)
    print("%s: FIR max|err| / max|y| = %.3g (rms %.3g), max|y| = %.4g" % (name, e.max() / np.abs(yo).max(), np.sqrt(np.mean(e ** 2)) / np.abs(yo).max(), np.abs(yo).max()))
    p = np.abs(yo[1:] * np.conj(yo[:-1]))
    ok = p >= 1e-3 * np.median(p)
    d = np.abs(np.angle(np.exp(1j * (am - ao))))
    print("   FM wrapped |dphi|: median %.3g, p99.9 %.3g, max over well-conditioned %.3g (%.2f %% masked), max over all %.3g" %
          (np.median(d), np.quantile(d, 0.999), d[ok].max(), 100 * (1 - ok.mean()), d.max()))


if __name__ == "__main__":
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 17)
    report("input B (FM tone + noise)", input_B(n))
    report("input A (iid u8 noise)", input_A(n))


def debug():
    x = input_B(1 << 14)
    yo, _ = oracle(x); ym, _ = model(x)
    e = np.abs(ym - yo) / np.abs(yo).max()
    bad = np.nonzero(e > 1e-5)[0]
    print(len(bad), bad[:40], e[bad[:10]])


def stopband(n, seed=5):
    """the failing GPU case: the carrier lands 62 kHz from the passband centre (f_off = -31 kHz on a +31 kHz carrier)"""
    import sys as _s, os as _o
    _s.path.insert(0, _o.path.dirname(_o.path.dirname(_o.path.dirname(_o.path.abspath(__file__)))))
    from oracle import dd_oracle as O
    return O.grid_c64(O.synth_iq_fm(n, FS, 2900, f_carrier=31000.0 + 25000.0 + 31000.0, f_mod=700.0, dev=4.0)).astype(np.complex128)
