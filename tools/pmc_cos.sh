#!/bin/bash
# SQ / LDS counters of the headline kernel (bench.py --steps 5, no side lines), separate --pmc passes; LIBV=build/variants/lib_N.so
# profiles a variant build instead (DD_LIB_PATH).   gpurun -- tools/pmc_cos.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
tag=${1:-pmc_cos}
[ -n "$LIBV" ] && export DD_LIB_PATH=$LIBV
i=0
: > gpurun_out/$tag.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE"; do
  rm -rf gpurun_out/${tag}_$i
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_$i -o p -- python3 bench.py --steps 5 --warmup 2 --ramp-ms 20 --no-cpu-baseline --no-side --steady-ms 300 > /dev/null 2> gpurun_out/${tag}_$i.err
  python3 tools/pmc_summary.py gpurun_out/${tag}_$i | grep -A6 "${PMC_KERNEL:-k_chain_cos1k}" >> gpurun_out/$tag.txt || tail -3 gpurun_out/${tag}_$i.err
  rm -rf gpurun_out/${tag}_$i
  i=$((i+1))
done
cat gpurun_out/$tag.txt
