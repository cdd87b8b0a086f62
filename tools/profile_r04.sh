#!/bin/bash
# round 4, end of round: the whole GPU suite, the bench line (with side configs and CPU baselines), the same command under rocprofv3
# --kernel-trace --stats, the PMC traffic passes of the bench's kernel, the NOAA path's stages and kernel statistics -- one gpurun call
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -6 | cut -c1-300 > gpurun_out/r04_final_tests.txt
cat gpurun_out/r04_final_tests.txt
python3 bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; tail -c 1500 gpurun_out/r04_bench.json | head -c 600; echo
bash tools/profile_bench.sh > gpurun_out/r04_profile_bench.txt 2>&1; head -8 gpurun_out/prof_kernel_stats.csv | cut -c1-220
bash tools/pmc_traffic.sh
python3 tools/bench_noaa.py 60 --stages > gpurun_out/r04_noaa_stages.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_noaa_stages.txt | tail -10
bash tools/profile_noaa.sh 60 > gpurun_out/r04_noaa_profile.txt 2>&1; head -12 gpurun_out/r04_noaa_profile.txt | cut -c1-200
bash tools/noaa_timeline.sh 60 > gpurun_out/r04_noaa_timeline.txt 2>&1; tail -3 gpurun_out/r04_noaa_timeline.txt
