#!/bin/bash
# round 4, GPU call 25: the three-launch envelope transform (dd_hconv_kernels.h): parity, then the NOAA stage timings
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_audio.py -x -q -k "sync_envelope or accurate_sync or c4_" 2>&1 | grep -v amdgpu.ids | tail -15
for m in own lib; do
  echo "== DD_SYNC_HILBERT=$m"
  if [ $m = lib ]; then export DD_SYNC_HILBERT=lib; fi
  timeout 300 python tools/bench_noaa.py 60 --stages 2>&1 | grep -v amdgpu.ids | tail -12
done
