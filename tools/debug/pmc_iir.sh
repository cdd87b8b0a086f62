#!/bin/bash
# SQ counters of the block passes of the IIR over 2^26 complex64 samples (dd_iir_c64), separate passes:  tools/debug/pmc_iir.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cat > /tmp/one_iir.py <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from directdemod_amd import _hip, filters
_hip.require_gpu()
n = 1 << 26
rng = np.random.default_rng(1)
x = (rng.standard_normal(n, dtype=np.float32) + 1j * rng.standard_normal(n, dtype=np.float32)).astype(np.complex64)
d = _hip.DevArray.from_host(x)
f = filters.butter(2048000, 20000.0, storeState=False)
for _ in range(2):
    y = f.applyOn(d); _hip.sync(); del y
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR"; do
  rm -rf gpurun_out/pmc_i$i
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_i$i -o p -- python3 /tmp/one_iir.py > /dev/null 2> gpurun_out/pmc_i$i.err
  python3 tools/pmc_summary.py gpurun_out/pmc_i$i | grep -A8 "k_iir_blocks_w32" || tail -3 gpurun_out/pmc_i$i.err
  i=$((i+1))
done
