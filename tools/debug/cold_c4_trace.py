#!/usr/bin/env python3
"""time line of the FIRST getCrudeSync / getAccurateSync of a fresh process (the shape of tools/bench_noaa_cold.py): start and end of the calls
underneath, per thread, ms from the start of getCrudeSync.   usage: python tools/debug/cold_c4_trace.py recording.npy"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from directdemod_amd import _hip, noaa_sync, source, _ops, comm
_hip.require_gpu()
raw = np.load(sys.argv[1])
log = []
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            log.append((label or name, t, time.perf_counter(), threading.current_thread().name))
    setattr(obj, name, g)
wrap(source._u8source, "read_device_raw")
wrap(_ops, "noaa_prepare")
wrap(_ops, "crude_tail")
wrap(noaa_sync.noaa_sync, "audio")
wrap(_hip, "wait_copy_warmup")
for n in ("sync_windows_multi", "noaa_sync_windows_multi", "sync_windows"):
    if hasattr(_ops, n):
        wrap(_ops, n)
wrap(comm, "_run_batch")
wrap(comm.commSignal, "_materialise", "commSignal._materialise")
src = source.IQarray(raw, 2048000)
obj = noaa_sync.noaa_sync(src, 30000.0)
_hip.sync()
t0 = time.perf_counter()
sa, sb = obj.getCrudeSync()
_hip.sync()
t1 = time.perf_counter()
acc = obj.getAccurateSync()
_hip.sync()
t2 = time.perf_counter()
print("first getCrudeSync %.2f ms, first getAccurateSync %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
for l, a, b, th in sorted(log, key=lambda r: r[1]):
    print("  %8.2f .. %8.2f  (%7.2f ms)  %-28s %s" % ((a - t0) * 1e3, (b - t0) * 1e3, (b - a) * 1e3, l, th))
