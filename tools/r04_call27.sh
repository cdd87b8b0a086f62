#!/bin/bash
# round 4, GPU call 27: correlation kernels (run table out of LDS; comb form for the accurate windows): parity, timings, timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_audio.py -x -q -k "sync or c4_ or crude or xcorr" 2>&1 | grep -v amdgpu.ids | tail -15
timeout 300 python tools/bench_noaa.py 60 --stages 2>&1 | grep -v amdgpu.ids | tail -12
tools/noaa_timeline.sh 60 > gpurun_out/r04_noaa_timeline2.txt 2>&1; grep -E "xcorr|span" gpurun_out/r04_noaa_timeline2.txt | tail -8
