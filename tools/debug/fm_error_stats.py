#!/usr/bin/env python3
"""FM angle error statistics of the M = 1 kernels against the float64 oracle on the bench input shape (2^19 samples):
   KERNELS=fft1k,ab python tools/debug/fm_error_stats.py"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from oracle import dd_oracle as O
from directdemod_amd import _hip, comm, filters, demod_fm
_hip.require_gpu()
fs, L = 2400000, 1 << 19
for seed, kind in ((1235, "fm"), (1234, "noise")):
    raw = O.synth_iq_fm(L, fs, seed, f_carrier=25e3, f_mod=1e3, dev=5.0) if kind == "fm" else O.synth_iq_noise(L, seed)
    x = O.grid_c64(raw)
    y = O.FilterState(O.win_hamming(255)).applyOn(O.nco(x, 25000.0, fs))
    ref, _ = O.fm_demod(y, None)
    mag = np.abs(y[1:] * np.conj(y[:-1]))
    for kern in os.environ.get("KERNELS", "fft1k,ab").split(","):
        _hip.select_kernel(kern)
        s = comm.commSignal(fs, x).offsetFreq(25000.0).filter(filters.hamming(255)).funcApply(demod_fm.demod_fm().demod)
        d = np.abs(np.angle(np.exp(1j * (s.signal - ref))))
        well = mag >= 0.1 * np.median(mag)
        print("%-5s %-5s max %.3g  max(well) %.3g  p99.9 %.3g  p99 %.3g  median %.3g  rms %.3g" % (kind, kern, d.max(), d[well].max(), np.percentile(d, 99.9), np.percentile(d, 99), np.median(d), np.sqrt(np.mean(d ** 2))))
