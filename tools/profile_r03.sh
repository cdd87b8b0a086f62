#!/bin/bash
# Round-3 evidence in one gpurun call -> gpurun_out/r03/ (copied into profiles/ by hand afterwards):
#   kernel stats of the default bench run (headline k_chain_fft1k + every side line), SQ / LDS counters and
#   FETCH_SIZE / WRITE_SIZE of the headline kernel and of the one-launch chunk-list kernel, micro-benchmarks.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/r03; rm -rf $O; mkdir -p $O
git rev-parse HEAD > $O/git_head 2>/dev/null || true
# 1. kernel stats: bench.py (headline + side lines), rocprofv3 --kernel-trace --stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
S=$(find $O/prof -name '*kernel_stats.csv' | head -1); python3 tools/trim_profile.py $S $O/bench_kernel_stats.csv; head -12 $O/bench_kernel_stats.csv | cut -c1-200
# 2. the plain bench line (not profiled), for the record
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
# 3. PMC, headline kernel: SQ sets, then FETCH / WRITE (separate passes)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_h$i -o p -- python3 bench.py --steps 5 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-side > /dev/null 2> $O/pmc_h$i.err
  python3 tools/pmc_summary.py $O/pmc_h$i | grep -A5 "k_chain_fft1k" >> $O/headline_pmc.txt
  i=$((i+1))
done
cat $O/headline_pmc.txt
# 4. PMC, the chunk-list kernel (C3 shape) and the decimating kernel one chunk: FETCH / WRITE + SQ
cat > /tmp/one_multi.py <<'PY'
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, scipy.signal as ss
from directdemod_amd import _hip
import bench
_hip.require_gpu()
lib = _hip.lib()
dev = torch.device("cuda", 0)
n = 1 << 26
x = bench.make_input(torch, n, 0, dev, 7)
out = torch.empty(n, dtype=torch.float32, device=dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rz = np.ascontiguousarray(ss.remez(127, [0, 100e3, 150e3, 4999999], [1, 0], fs=1e7))
h = C.c_void_p()
_hip.check(lib.dd_chain_create(C.byref(h), rz.ctypes.data_as(C.POINTER(C.c_double)), 127, _hip.cycles_q64(250000.0, 10000000), 50, _hip.DD_CHAIN_NCO | _hip.DD_CHAIN_FM), "create")
nch = 16
cb = (C.c_int64 * (nch + 1))(*[i << 22 for i in range(nch + 1)])
cn = (C.c_int64 * nch)()
got = C.c_int64(0)
for _ in range(4):
    lib.dd_chain_reset(h, stream)
    _hip.check(lib.dd_chain_process_chunks(h, x.data_ptr(), out.data_ptr(), cb, nch, cn, stream), "chunks")
    lib.dd_chain_reset(h, stream)
    _hip.check(lib.dd_chain_process(h, x.data_ptr(), out.data_ptr(), n, C.byref(got), stream), "process")
torch.cuda.synchronize()
PY
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"; do
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_m$i -o p -- python3 /tmp/one_multi.py > /dev/null 2> $O/pmc_m$i.err
  python3 tools/pmc_summary.py $O/pmc_m$i | grep -A5 "k_chain_decim" >> $O/decim_pmc.txt
  i=$((i+1))
done
cat $O/decim_pmc.txt
# 5. side benchmarks and micro-benchmarks
( echo "== tools/fft_ab.py (ab vs fft1k, complex64 and u8 input, tap classes)"; KERNELS=ab,fft1k NTAPS=255,151,63 REPS=150 ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps; U8=1 KERNELS=ab,fft1k NTAPS=255,151,63 REPS=150 ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps
  echo "== tools/bench_noaa.py 120 --stages"; python3 tools/bench_noaa.py 120 --stages 2>&1 | grep -v amdgpu.ids
  echo "== tools/ubench/bin/mfma_i8_vs_f16"; tools/ubench/bin/mfma_i8_vs_f16
  echo "== tools/ubench/bin/pk_chain"; tools/ubench/bin/pk_chain
  echo "== tools/ubench/bin/vmem_issue"; tools/ubench/bin/vmem_issue ) > $O/side_benchmarks.txt 2>&1
tail -60 $O/side_benchmarks.txt
# 6. ablations of k_chain_fft1k (build/variants/lib_*.so, built in the container)
if ls build/variants/lib_*.so > /dev/null 2>&1; then KERNELS=fft1k ROUNDS_V=1 bash tools/variants_fft.sh 2>&1 | grep -v amdgpu.ids > $O/fft1k_ablations.txt; cat $O/fft1k_ablations.txt; fi
rm -rf $O/prof $O/pmc_h* $O/pmc_m*
