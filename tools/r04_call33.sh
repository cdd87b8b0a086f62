#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_audio.py -x -q -k "crude or c4_ or xcorr_and_peaks" 2>&1 | grep -v amdgpu.ids | tail -6
DD_CRUDE_TRACE=1 timeout 300 python tools/debug/crude_graph.py 8 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python tools/bench_noaa.py 60 2>&1 | grep -v amdgpu.ids | tail -1
