#!/bin/bash
# round 4, GPU call 30: k_xcorr_runs_pk with two runs' look-ups in flight (-DDD_XC_UNROLL2, build/variants/lib_1.so) against the product
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
for v in orig 1 orig 1; do
  if [ $v = orig ]; then cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so; else cp build/variants/lib_$v.so directdemod_amd/libdirectdemod_hip.so; fi
  echo "== lib $v"
  tools/noaa_timeline.sh 60 2>&1 | grep -E "k_xcorr_runs_pk|span" | tail -5
done
cp build/variants/lib_1.so directdemod_amd/libdirectdemod_hip.so
timeout 600 python -m pytest tests/test_gpu_audio.py -x -q -k "accurate or c4_" 2>&1 | grep -v amdgpu.ids | tail -3
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
