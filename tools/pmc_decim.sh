#!/bin/bash
# SQ counters of the decimating kernel, separate passes:  CASE=C4|C3|C4u8 tools/pmc_decim.sh   (round 6: CASE, the matrix-instruction counters)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
CASE=${CASE:-C4}
cat > /tmp/one_decim.py <<'PY'
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, scipy.signal
from directdemod_amd import _hip as hip
import bench
hip.require_gpu()
lib = hip.lib()
dev = torch.device("cuda", 0)
n = 1 << 26
case = os.environ.get("CASE", "C4")
x = bench.make_input(torch, n, 0, dev, 7)
fl = 0
if case == "C3":
    taps, M, f, fs = scipy.signal.remez(127, [0, 100e3, 150e3, 4999999], [1, 0], fs=1e7), 50, 250000.0, 1e7
else:
    taps, M, f, fs = scipy.signal.windows.blackmanharris(151), 34, 30000.0, 2048000.0
    if case == "C4u8":
        x = (x + 127.5).round().clamp(0, 255).to(torch.uint8).contiguous()
        fl = hip.DD_CHAIN_U8_INPUT
taps = np.ascontiguousarray(taps, dtype=np.float64)
h = C.c_void_p()
hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), len(taps), hip.cycles_q64(f, fs), M, hip.DD_CHAIN_NCO | hip.DD_CHAIN_FM | fl))
out = torch.empty(n // M + 8, dtype=torch.float32, device=dev)
for _ in range(4):
    lib.dd_chain_reset(h, None)
    hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, None, None))
torch.cuda.synchronize()
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC"; do
  rm -rf gpurun_out/pmc_d$i
  CASE=$CASE rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_d$i -o p -- python3 /tmp/one_decim.py > /dev/null 2> gpurun_out/pmc_d$i.err
  python3 tools/pmc_summary.py gpurun_out/pmc_d$i | grep -A8 "k_chain_decim" || tail -3 gpurun_out/pmc_d$i.err
  i=$((i+1))
done
