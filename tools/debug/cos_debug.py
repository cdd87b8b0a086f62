#!/usr/bin/env python3
"""where k_chain_cos1k's angles leave the float64 oracle's: positions (relative to the chunk cuts and to the 1024-sample rows) of the worst outputs"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from oracle import dd_oracle as O
from directdemod_amd import _hip, comm, filters, demod_fm, chunker
_hip.require_gpu()
f_off = float(os.environ.get("F_OFF", "-31000"))
fs = 2400000
cuts = np.cumsum([0, 1, 2, 253, 1023, 1024, 1025, 2047, 2048, 5000, 3, 70001, 777, 4096 * 9 + 5])
L = int(cuts[-1])
raw = O.synth_iq_fm(L, fs, 2900, f_carrier=abs(f_off) if f_off else 1000.0, f_mod=700.0, dev=4.0)
x = O.grid_c64(raw)
taps = O.win_hamming(255)
class S_:
    length = L
for kern in ("cos1k", "fft1k"):
    _hip.select_kernel(kern)
    flt = filters.hamming(255); fm = demod_fm.demod_fm(); ck = chunker.chunker(S_()); out = comm.commSignal(fs); fo = O.FilterState(taps)
    last, idx, refs, mags = None, 0, [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        s = comm.commSignal(fs, x[a:b], ck)
        if f_off: s.offsetFreq(f_off)
        s.filter(flt).funcApply(fm.demod); out.extend(s)
        y = fo.applyOn(O.nco(x[a:b], f_off, fs, idx) if f_off else x[a:b]); idx += b - a
        prv = last; r, last = O.fm_demod(y, last); refs.append(r)
        yy = y if prv is None else np.concatenate([[prv], y]); mags.append(np.abs(yy[1:] * np.conj(yy[:-1])))
    ref = np.concatenate(refs); mag = np.concatenate(mags)
    got = np.asarray(out.signal, dtype=np.float64)
    d = np.abs(np.angle(np.exp(1j * (got - ref))))
    ok = mag >= 1e-3 * np.median(mag)
    print(kern, "kernel id", flt._last_kernel(), "median %.3g max(masked) %.3g max(all) %.3g" % (np.median(d), d[ok].max(), d.max()))
    worst = np.argsort(-(d * ok))[:12]
    for w in sorted(worst):
        c = np.searchsorted(cuts, w + 1, side="right") - 1
        print("   out %7d (sample %7d): chunk %2d offset %6d of %6d, err %.3g, |z|/median %.3g" % (w, w + 1, c, w + 1 - cuts[c], cuts[c + 1] - cuts[c], d[w], mag[w] / np.median(mag)))
