"""the first C4 call of a fresh process as bench.py starts it: a CHILD of a process that holds a GPU context (torch) and 1 GB of HBM, the recording in /dev/shm"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
x = torch.zeros(1 << 28, dtype=torch.float32, device="cuda:0"); torch.cuda.synchronize()
raw = bench.synth_apt_iq(60.0, 2048000, seed=1)
fd, path = tempfile.mkstemp(suffix=".npy", dir="/dev/shm"); os.close(fd); np.save(path, raw)
env = dict(os.environ, DD_CRUDE_TRACE="1", DD_SYNC_TRACE="1")
for i in range(3):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "debug", "cold_c4_trace.py"), path], capture_output=True, text=True, env=env)
    print("\n".join(l for l in (r.stdout + r.stderr).splitlines() if "read_device_raw" not in l and "amdgpu.ids" not in l and "needle " not in l)); print("----")
for i in range(3):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_noaa_cold.py"), path], capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1][:300])
os.unlink(path)
