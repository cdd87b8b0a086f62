#!/bin/bash
# memory-side SQ counters of k_chain_fft
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export KERNELS=${KERNELS:-fft} REPS=${REPS:-10} ROUNDS=1
i=0
for set in "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_IFETCH" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU"; do
  rm -rf gpurun_out/pmc_g$i
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_g$i -o p -- python3 tools/fft_ab.py > /dev/null 2> gpurun_out/pmc_g$i.err
  python3 tools/pmc_summary.py gpurun_out/pmc_g$i | grep -A6 "${PMC_KERNEL:-k_chain_fft}" || tail -3 gpurun_out/pmc_g$i.err
  i=$((i+1))
done
