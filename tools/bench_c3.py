#!/usr/bin/env python3
"""Side measurement: BASELINE config 3 through the drop-in classes, chunk loop as in decode_fm.getAudio
(decode_fm.py:42-72): 2^26 complex64 samples @10 MS/s, device resident, 16 chunks of 2^22, offsetFreq 250 kHz,
remez(127), bwLim 200 kS/s (/50), FM, strict bwLim to 11 025 S/s, extend."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from directdemod_amd import _hip, comm, filters, demod_fm, chunker
_hip.require_gpu()
fs, n, chunk = 10000000, 1 << 26, 1 << 22
rng = np.random.default_rng(2235)
x = (np.clip(np.round(60 * np.exp(2j * np.pi * 250e3 * np.arange(chunk) / fs) + 4 * (rng.standard_normal(chunk) + 1j * rng.standard_normal(chunk)) + 127.5 * (1 + 1j)), 0, 255) - 127.5 * (1 + 1j)).astype(np.complex64)
d = _hip.DevArray.from_host(np.tile(x, n // chunk))


class Src:
    length = n
    sampFreq = fs


for rep in range(3):
    flt = filters.remez(fs, [[0, 100e3], [150e3, 4999999]], [1, 0], ntaps=127)
    fm = demod_fm.demod_fm()
    ck = chunker.chunker(Src(), chunk)
    audio = comm.commSignal(11025)
    _hip.sync()
    t0 = time.perf_counter()
    for a, b in ck.getChunks:
        sig = comm.commSignal(fs, d.view(a, b - a), ck).offsetFreq(250000.0).filter(flt).bwLim(200000, uniq="First") \
            .funcApply(fm.demod).bwLim(11025, True)
        audio.extend(sig)
    out = audio.device_signal
    _hip.sync()
    dt = time.perf_counter() - t0
    print("run %d: C3 chunk loop (16 chunks): %.2f ms total = %.1f GS/s in, %d audio samples out" % (rep, dt * 1e3, n / dt / 1e9, audio.length))
