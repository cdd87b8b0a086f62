import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from directdemod_amd import _hip, noaa_sync, source
from oracle import dd_oracle as O
_hip.require_gpu()
raw = O.synth_apt_iq(60.0, seed=1)
src = source.IQarray(raw, 2048000)
o = noaa_sync.noaa_sync(src, 30000.0); o.getCrudeSync(); o.getAccurateSync()
for _ in range(4):
    _hip.sync(); t0 = time.perf_counter(); o.getAccurateSync(); _hip.sync(); print("getAccurateSync %.3f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
