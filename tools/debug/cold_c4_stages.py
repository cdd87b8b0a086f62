#!/usr/bin/env python3
"""where the FIRST getCrudeSync of a fresh process spends its time: stage by stage, each synchronised (recording from a .npy of raw u8 pairs)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
from directdemod_amd import _hip, noaa_sync, source, constants, _ops
_hip.require_gpu()
raw = np.load(sys.argv[1])
marks = [("import + gpu init + np.load", time.perf_counter() - t0)]
def T(label, fn):
    _hip.sync(); t = time.perf_counter(); r = fn(); _hip.sync(); marks.append((label, time.perf_counter() - t)); return r
src = T("source.IQarray", lambda: source.IQarray(raw, 2048000))
obj = T("noaa_sync object", lambda: noaa_sync.noaa_sync(src, 30000.0))
aud = T("audio (upload + fused chunk-list launch)", lambda: obj.audio(constants.NOAA_CRUDESYNCSAMPRATE, False))
needles = [noaa_sync.sync_needle(constants.NOAA_SYNCA, aud.sampRate), noaa_sync.sync_needle(constants.NOAA_SYNCB, aud.sampRate)]
T("crude tail, first call", lambda: _ops.crude_tail(aud.device_signal, aud.sampRate, needles))
T("crude tail, second call", lambda: _ops.crude_tail(aud.device_signal, aud.sampRate, needles))
aud = T("audio again", lambda: obj.audio(constants.NOAA_CRUDESYNCSAMPRATE, False))
for l, v in marks:
    print("  %-44s %9.2f ms" % (l, v * 1e3))
