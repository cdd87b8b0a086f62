#!/bin/bash
# round 4, GPU call 31: extend() of a finished chunk list adopts the contiguous outputs (no per-chunk copies): class-level tests, C3 / C4 wall
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python -m pytest tests/test_gpu_audio.py tests/test_gpu_parity.py -x -q -k "class or chunk or extend or c3_ or c4_ or lazy or comm" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python tools/bench_noaa.py 60 2>&1 | grep -v amdgpu.ids | tail -1
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for s in d['extra']['side']:
    if s['config'].startswith(('C3 end', 'C4 end')):
        print(s['config'][:40], {k: v for k, v in s.items() if 'ms' in k})
"
