#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -12 | cut -c1-300 > gpurun_out/r04_call10_tests.txt
cat gpurun_out/r04_call10_tests.txt
python3 tools/bench_noaa.py 60 --stages > gpurun_out/r04_noaa_stages.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_noaa_stages.txt
bash tools/profile_noaa.sh 60 > gpurun_out/r04_noaa_profile.txt 2>&1; cat gpurun_out/r04_noaa_profile.txt | cut -c1-200 | head -42
