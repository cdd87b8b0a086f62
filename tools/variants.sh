#!/bin/bash
# run every build/variants/lib_N.so through parity, the stamps and the steady-state bench
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt
for f in build/variants/lib_*.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  echo "== $f"
  [ -z "$NO_PARITY" ] && python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
  DD_STAMPS=3 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --ramp-ms 0 2>&1 | grep -i "stamps" | sed -n ${STAMP_LINES:-4,8}p
  for k in 1 2; do python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
