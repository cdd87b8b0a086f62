#!/bin/bash
# round 5: the PMC traffic passes and the clock / power record of the headline kernel first (so that the bench line quotes records of THIS
# tree), then the bench line (side configs and CPU baselines), the same command under rocprofv3 --kernel-trace --stats, the SQ passes, the
# arithmetic-only micro-benchmark, the NOAA stage table -- one gpurun call:
#   gpurun -- "DD_GIT_HEAD=$(git rev-parse --short HEAD) tools/profile_r05.sh"   then copy gpurun_out/r05_* and the two .json files into profiles/
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/pmc_traffic.sh | tail -1 | cut -c1-400
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
python3 tools/power_json.py gpurun_out/power.json | cut -c1-900
cp gpurun_out/power.json profiles/power.json
python3 bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; tail -c 2500 gpurun_out/r05_bench.json | head -c 900; echo
bash tools/profile_bench.sh > gpurun_out/r05_profile_bench.txt 2>&1; cp gpurun_out/prof_kernel_stats.csv gpurun_out/r05_bench_kernel_stats.csv; cp gpurun_out/prof_bench.json gpurun_out/r05_bench_profiled.json; head -6 gpurun_out/prof_kernel_stats.csv | cut -c1-200
bash tools/pmc_cos.sh r05_cos1k_pmc > /dev/null 2>&1; cat gpurun_out/r05_cos1k_pmc.txt
SKIP_FFT=1 bash tools/ubench/run_cosfir.sh > /dev/null 2>&1; cp gpurun_out/r05_cosfir.txt gpurun_out/r05_cosfir_ubench.txt
python3 tools/bench_noaa.py 60 --stages > gpurun_out/r05_noaa_stages.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05_noaa_stages.txt | tail -10
