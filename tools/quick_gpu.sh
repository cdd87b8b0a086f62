#!/bin/bash
# quick GPU check used while tuning the MFMA kernel: parity tests, 3 bench runs, stage stamps
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_events'], d['roofline']['frac'])"; done
DD_STAMPS=1 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep -i "stamps" | head -20
