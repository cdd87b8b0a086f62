#!/usr/bin/env python3
"""Where the host time of getCrudeSync / getAccurateSync goes (cProfile over repeated calls on a resident 60 s recording)."""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from directdemod_amd import _hip, noaa_sync, source
import bench
_hip.require_gpu()
dur = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
raw = bench.synth_apt_iq(dur, 2048000, seed=1)
src = source.IQarray(raw, 2048000)
for _ in range(3):
    o = noaa_sync.noaa_sync(src, 30000.0); o.getCrudeSync(); o.getAccurateSync()
def crude():
    for _ in range(20):
        o = noaa_sync.noaa_sync(src, 30000.0); o.getCrudeSync(); _hip.sync()
def acc():
    o = noaa_sync.noaa_sync(src, 30000.0); o.getCrudeSync()
    for _ in range(10):
        o.getAccurateSync(); _hip.sync()
for name, fn in (("crude x20", crude), ("accurate x10", acc)):
    t0 = time.perf_counter(); pr = cProfile.Profile(); pr.enable(); fn(); pr.disable()
    print("==", name, "%.2f ms total" % ((time.perf_counter() - t0) * 1e3))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
for b in (64, 128, 256):
    os.environ["DD_SYNC_BATCH"] = str(b)
    o = noaa_sync.noaa_sync(src, 30000.0); o.getCrudeSync(); o.getAccurateSync(); _hip.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); o.getAccurateSync(); _hip.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print("DD_SYNC_BATCH=%d: accurate sync %.2f ms (min of 5)" % (b, min(ts)))
