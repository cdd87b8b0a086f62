#!/bin/bash
# does the kernel time depend on how long the GPU has been busy? (DVFS ramp / power cap)
for cfg in "20 3" "200 50" "2000 200" "2000 2000"; do set -- $cfg
python bench.py --no-cpu-baseline --steps $1 --warmup $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps',d['steps'],'warmup',d['warmup'],d['value'], d['ms_per_step'], d['roofline']['kernel_ms_events'], d['roofline']['frac'])"; done
