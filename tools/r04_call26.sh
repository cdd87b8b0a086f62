#!/bin/bash
# round 4, GPU call 26: crude tail with one pinned copy back, both accurate-sync searches in one call: parity, then timings
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_audio.py -x -q -k "sync or c4_ or crude" 2>&1 | grep -v amdgpu.ids | tail -15
timeout 300 python tools/bench_noaa.py 60 --stages 2>&1 | grep -v amdgpu.ids | tail -12
