python - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import bench
np.save("/tmp/apt60.npy", bench.synth_apt_iq(60.0, 2048000, seed=1))
PY
for i in 1 2 3; do DD_CRUDE_TRACE=1 DD_SYNC_TRACE=1 python tools/debug/cold_c4_trace.py /tmp/apt60.npy 2>&1 | grep -v amdgpu.ids | cut -c1-300; echo ----; done
