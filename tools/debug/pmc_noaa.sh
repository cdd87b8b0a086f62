#!/bin/bash
# SQ counters of the accurate-sync kernels of config 4 (tools/bench_noaa.py 60), separate passes:  tools/debug/pmc_noaa.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  rm -rf gpurun_out/pmc_n$i
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_n$i -o p -- python3 tools/bench_noaa.py 60 > /dev/null 2> gpurun_out/pmc_n$i.err
  python3 tools/pmc_summary.py gpurun_out/pmc_n$i | grep -A4 "k_filtfilt_tile\|k_hc_\|k_filtfilt_cos\|k_xcorr_runs_pk\|k_scan_final(\|k_scan_part" || tail -3 gpurun_out/pmc_n$i.err
  i=$((i+1))
done
