#!/bin/bash
# SQ / LDS counters of the overlap-save FFT kernel (tools/fft_ab.py, KERNELS=fft), separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export KERNELS=${KERNELS:-fft} REPS=${REPS:-10} ROUNDS=1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE"; do
  rm -rf gpurun_out/pmc_f$i
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_f$i -o p -- python3 tools/fft_ab.py > /dev/null 2> gpurun_out/pmc_f$i.err
  python3 tools/pmc_summary.py gpurun_out/pmc_f$i | grep -A6 "${PMC_KERNEL:-k_chain_fft}" || tail -3 gpurun_out/pmc_f$i.err
  i=$((i+1))
done
