#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_audio.py -m gpu -q -x 2>&1 | tail -30 | cut -c1-300 > gpurun_out/r04_call7_tests.txt
cat gpurun_out/r04_call7_tests.txt
python3 tools/bench_noaa.py 60 --stages > gpurun_out/r04_noaa_stages.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_noaa_stages.txt
DD_SYNC_DIRECT_FIR=1 python3 tools/bench_noaa.py 60 2>&1 | grep resident
bash tools/profile_noaa.sh 60 > gpurun_out/r04_noaa_profile.txt 2>&1; cat gpurun_out/r04_noaa_profile.txt | cut -c1-200
