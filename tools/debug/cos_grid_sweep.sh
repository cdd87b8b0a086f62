for g in 512 511 509 500 480 448 400 384 768 765; do DD_COS_GRID=$g python bench.py --no-cpu-baseline --no-side --steady-ms 300 --steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('grid $g', d['config']['kernel'], 'kernel_ms', d['roofline']['kernel_ms'], 'steady', d['extra']['steady_check']['kernel_ms'], 'frac', d['roofline']['frac'])"; done
