#!/bin/bash
# round 6: the end-of-round record in one gpurun call -- the traffic passes of the decimating front ends and the clock / power record first (so that
# the bench line quotes records of THIS tree), then the bench line (side configs and CPU baselines), the same command under rocprofv3
# --kernel-trace --stats, the SQ counters of k_chain_decim_b, the driver's GPU test tier:
#   gpurun -- "GIT_REV=$(git rev-parse --short HEAD) DD_GIT_HEAD=$(git rev-parse --short HEAD) tools/profile_r06.sh"
# then copy gpurun_out/r06_* and the two .json files into profiles/
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/pmc_decimw_traffic.sh 2>&1 | grep -v amdgpu.ids | tail -4
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
python3 tools/power_json.py gpurun_out/power.json | cut -c1-600
cp gpurun_out/power.json profiles/power.json
python3 bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; tail -c 2500 gpurun_out/r06_bench.json | head -c 600; echo
bash tools/profile_bench.sh > gpurun_out/r06_profile_bench.txt 2>&1; cp gpurun_out/prof_kernel_stats.csv gpurun_out/r06_bench_kernel_stats.csv; cp gpurun_out/prof_bench.json gpurun_out/r06_bench_profiled.json; head -8 gpurun_out/prof_kernel_stats.csv | cut -c1-220
for c in C4 C3 C4u8; do echo "== $c"; CASE=$c bash tools/pmc_decim.sh 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06_decimb_pmc.txt 2>&1; tail -12 gpurun_out/r06_decimb_pmc.txt
python3 tools/bench_noaa.py 60 --stages > gpurun_out/r06_noaa_stages.txt 2>&1; grep -v amdgpu.ids gpurun_out/r06_noaa_stages.txt | tail -8
bash tools/run_gpu_tests.sh r06 > /dev/null 2>&1; tail -3 gpurun_out/r06_tests.txt; cp gpurun_out/r06_tests.txt gpurun_out/r06_gpu_tests.txt
