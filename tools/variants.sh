#!/bin/bash
# run every build/variants/lib_N.so through the stamps + steady-state bench
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt
for f in build/variants/lib_*.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  echo "== $f"
  DD_STAMPS=1 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep -i "stamps" | sed -n 4,20p
  python bench.py --no-cpu-baseline --steps 500 --warmup 200 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
