#!/bin/bash
# SQ counters of the decimating kernel (C4 front end shape), separate passes
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cat > /tmp/one_decim.py <<'PY'
import ctypes as C, os, sys
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
for _ in range(4):
    _hip.lib().dd_chain_reset(eng.h, stream)
    eng.process(x.data_ptr(), out.data_ptr(), n)
torch.cuda.synchronize()
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  rm -rf gpurun_out/pmc_d$i
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_d$i -o p -- python3 /tmp/one_decim.py > /dev/null 2> gpurun_out/pmc_d$i.err
  python3 tools/pmc_summary.py gpurun_out/pmc_d$i | grep -A8 "k_chain_decim"
  i=$((i+1))
done
