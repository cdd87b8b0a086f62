#!/usr/bin/env python3
"""dd_noaa_crude_tail N times on the audio of a resident 60 s recording: the same peaks on every call, and what a call costs
(DD_CRUDE_TRACE=1: host-side time stamps inside the call -- enqueueing done, synchronised, candidates per needle, grouping done).
usage: [DD_CRUDE_TRACE=1] python tools/debug/crude_tail_repeat.py [N]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from directdemod_amd import _hip, _ops, noaa_sync, source, constants
from oracle import dd_oracle as O      # synthetic generator only
_hip.require_gpu()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
raw = O.synth_apt_iq(60.0, seed=1)
src = source.IQarray(raw, 2048000)
obj = noaa_sync.noaa_sync(src, 30000.0)
aud = obj.audio(constants.NOAA_CRUDESYNCSAMPRATE, False)
d = aud.device_signal
needles = [noaa_sync.sync_needle(constants.NOAA_SYNCA, aud.sampRate), noaa_sync.sync_needle(constants.NOAA_SYNCB, aud.sampRate)]
ref = None
ts = []
for i in range(N):
    _hip.sync()
    t0 = time.perf_counter()
    res = _ops.crude_tail(d, aud.sampRate, needles)
    ts.append(time.perf_counter() - t0)
    assert res is not None
    pa, pb = res[0]
    if ref is None:
        ref = (pa.copy(), pb.copy())
    assert np.array_equal(pa, ref[0]) and np.array_equal(pb, ref[1]), "call %d differs" % i
ts = np.array(ts[5:]) * 1e3
print("%d calls, identical peaks (%d + %d); per call min %.3f median %.3f ms; checksum %d"
      % (N, len(ref[0]), len(ref[1]), ts.min(), np.median(ts), int(ref[0].sum() + ref[1].sum())))
