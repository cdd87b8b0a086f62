#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_audio.py -m gpu -q -x 2>&1 | tail -5 | cut -c1-300
python3 tools/bench_noaa.py 60 --stages 2>&1 | grep -v amdgpu.ids | tail -10
DD_CRUDE_GRAPH=0 python3 tools/bench_noaa.py 60 --stages 2>&1 | grep -v amdgpu.ids | grep "resident\|crude tail"
