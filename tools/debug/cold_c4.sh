#!/bin/bash
# the first C4 call of a fresh process (tools/bench_noaa_cold.py) on a 60 s synthetic recording, three fresh processes; DD_CRUDE_TRACE / DD_SYNC_TRACE show where it goes
python - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import bench
np.save("/tmp/apt60.npy", bench.synth_apt_iq(60.0, 2048000, seed=1))
PY
for i in 1 2 3; do DD_CRUDE_TRACE=1 DD_SYNC_TRACE=1 python tools/bench_noaa_cold.py /tmp/apt60.npy 2>&1 | grep -v amdgpu.ids | cut -c1-400; done
DD_AM_HILBERT=lib python tools/bench_noaa_cold.py /tmp/apt60.npy 2>&1 | grep "^{" | cut -c1-300
