#!/bin/bash
# SQ / LDS counters of the headline kernel (k_chain_mfma_ws, C2 workload), separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  rm -rf gpurun_out/pmc_w$i
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_w$i -o p -- python3 bench.py --no-cpu-baseline --no-side --steps 5 --warmup 2 --ramp-ms 0 > /dev/null 2> gpurun_out/pmc_w$i.err
  python3 tools/pmc_summary.py gpurun_out/pmc_w$i | grep -A6 "k_chain_mfma_" || tail -3 gpurun_out/pmc_w$i.err
  i=$((i+1))
done
