#!/bin/bash
# soak: many launches of the persistent kernels (barrier / LDS-counter hand-overs), bounded by timeouts
timeout 120 python bench.py --no-cpu-baseline --steps 20000 --warmup 10 --ramp-ms 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mfma ws 20000 steps:', d['value'], d['ms_per_step'], d['extra']['output_rms_rad'])"
timeout 120 python - <<'PY'
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from directdemod_amd import _hip, shard
import bench
_hip.require_gpu()
dev = torch.device("cuda", 0)
n = 1 << 26
x = bench.make_input(torch, n, 0, dev, 7)
out = torch.empty(n, dtype=torch.float32, device=dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
k = np.arange(151)
bh = 0.35875 - 0.48829 * np.cos(2 * np.pi * k / 150) + 0.14128 * np.cos(4 * np.pi * k / 150) - 0.01168 * np.cos(6 * np.pi * k / 150)
eng = shard.HipChainEngine(bh, 30000.0, 2048000, 34, stream=stream)
ref = None
t0 = time.time()
for i in range(5000):
    _hip.lib().dd_chain_reset(eng.h, stream)
    eng.process(x.data_ptr(), out.data_ptr(), n)
    if i % 1000 == 0:
        torch.cuda.synchronize()
        s = float(out[:1000000].double().sum())
        assert ref is None or s == ref, (i, s, ref)
        ref = s
torch.cuda.synchronize()
print("decim persistent 5000 launches ok, checksum stable, %.1f s" % (time.time() - t0))
PY
