#!/bin/bash
# rows per run of k_chain_cos1k's wave -> row map (DD_COS_RUN; 0 = one contiguous run per wave), product library and the memory-only build
for lib in "" build/variants/lib_1.so; do
  for r in 0 2 4 8 16; do
    DD_LIB_PATH=$lib DD_COS_RUN=$r python bench.py --no-cpu-baseline --no-side --steady-ms 300 --steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib:-product} run $r', 'kernel_ms', d['roofline']['kernel_ms'], 'steady', d['extra']['steady_check']['kernel_ms'], 'frac', d['roofline']['frac'], 'rms', round(d['extra']['output_rms_rad'], 5))"
  done
done
