#!/bin/bash
# round 4, GPU call 28: the generalised float64 transform (rows of 256 / 512; chirp-z resampler through it): parity, C3 / C4 timings
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_audio.py -x -q -k "sync or c4_ or crude or resample or c3_ or class_chunk" 2>&1 | grep -v amdgpu.ids | tail -12
timeout 300 python tools/bench_noaa.py 60 2>&1 | grep -v amdgpu.ids | tail -1
for own in 1 0; do
  echo "== DD_CZT_OWN=$own"
  DD_CZT_OWN=$own timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for s in d['extra']['side']:
    if s['config'].startswith('C3 end'):
        print({k: v for k, v in s.items() if 'ms' in k})
"
done
