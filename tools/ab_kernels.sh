#!/bin/bash
# the headline step with each M = 1 kernel in ONE gpurun call (same box): ms per step, roofline fraction
#   gpurun -- tools/ab_kernels.sh [kernels...]        default: auto fft1k
for k in ${@:-auto fft1k}; do
  for r in 1 2; do
    DD_MFMA_KERNEL=$k python bench.py --no-cpu-baseline --no-side --steady-ms 300 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$k', d['config']['kernel'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms_events'], 'frac', d['roofline']['frac'], 'steady', d['extra']['steady_check']['kernel_ms'], 'rms', round(d['extra']['output_rms_rad'], 5))"
  done
done
