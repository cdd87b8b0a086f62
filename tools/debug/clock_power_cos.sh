#!/bin/bash
# shader clock and package power (rocm-smi, read-only) while one kernel loops: k_chain_cos1k, k_chain_fft1k and every build/variants/lib_N.so (cos1k)
DUR=4 KERNEL=cos1k python3 tools/debug/clock_power.py 2>&1 | python3 tools/debug/clock_power_summ.py
DUR=4 KERNEL=fft1k python3 tools/debug/clock_power.py 2>&1 | python3 tools/debug/clock_power_summ.py
for f in build/variants/lib_*.so; do
  [ -f "$f" ] || continue
  i=$(basename $f .so | sed 's/lib_//'); echo "variant $(sed -n ${i}p build/variants/index.txt)"
  DUR=4 KERNEL=cos1k LIB=$f python3 tools/debug/clock_power.py 2>&1 | python3 tools/debug/clock_power_summ.py
done
