#!/usr/bin/env python3
"""Side measurement: config 4 (NOAA APT sync detection) end to end on the device -- crude sync over the
whole recording and the accurate-sync windows -- on a synthetic recording (not the headline metric)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from directdemod_amd import _hip, noaa_sync, source
from oracle import dd_oracle as O      # synthetic generator only
_hip.require_gpu()
dur = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
raw = O.synth_apt_iq(dur, seed=1)
def run(src, label):
    obj = noaa_sync.noaa_sync(src, 30000.0)
    _hip.sync()
    t0 = time.perf_counter()
    sa, sb = obj.getCrudeSync()
    _hip.sync()
    t1 = time.perf_counter()
    acc = obj.getAccurateSync()
    _hip.sync()
    t2 = time.perf_counter()
    nwin = len(acc[0][0]) + len(acc[1][0])
    print("%s: %.0f s recording (%d IQ samples): crude sync %.1f ms (%d + %d syncs), accurate sync %.1f ms for %d windows = %.3f ms/window"
          % (label, dur, src.length, (t1 - t0) * 1e3, len(sa), len(sb), (t2 - t1) * 1e3, nwin, (t2 - t1) * 1e3 / max(1, nwin)))
    return sa


run(source.IQarray(raw, 2048000), "warm-up (plans, first launches)")
src = source.IQarray(raw, 2048000)
run(src, "recording on the host (uploaded during the crude pass)")
sa = run(src, "recording resident in HBM")

if "--stages" in sys.argv:
    from directdemod_amd import constants
    def T(label, fn):
        _hip.sync(); t = time.perf_counter(); r = fn(); _hip.sync()
        print("  %-34s %7.2f ms" % (label, (time.perf_counter() - t) * 1e3)); return r
    obj = noaa_sync.noaa_sync(src, 30000.0)
    aud = T("audio (fused chunk-list launch)", lambda: obj.audio(constants.NOAA_CRUDESYNCSAMPRATE, False))
    from directdemod_amd import _ops
    needles = [noaa_sync.sync_needle(constants.NOAA_SYNCA, aud.sampRate), noaa_sync.sync_needle(constants.NOAA_SYNCB, aud.sampRate)]
    T("crude tail, ONE call (fused)", lambda: _ops.crude_tail(aud.device_signal, aud.sampRate, needles))
    T("crude tail, ONE call (again)", lambda: _ops.crude_tail(aud.device_signal, aud.sampRate, needles))
    env = T("envelope (240000-blocks)", lambda: obj.envelope(aud))
    T("correlate + peaks, sync A", lambda: obj.correlate_and_find_peaks(env, constants.NOAA_SYNCA))
    T("correlate + peaks, sync B", lambda: obj.correlate_and_find_peaks(env, constants.NOAA_SYNCB))
    width = int(3 * constants.NOAA_T * len(constants.NOAA_SYNCA) * src.sampFreq)
    starts = [int(c) - width for c in sa / env.sampRate * src.sampFreq if int(c) - width >= 0 and int(c) + width <= src.length]
    T("accurate windows, sync A (%d)" % len(starts), lambda: obj.accurate_windows(starts, 2 * width, constants.NOAA_SYNCA))
