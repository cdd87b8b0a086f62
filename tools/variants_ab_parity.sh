#!/bin/bash
# like variants_ab.sh, with the MFMA parity tests for every variant first
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt
for f in /tmp/lib_orig.so build/variants/lib_*.so; do
  [ "$f" != /tmp/lib_orig.so ] && cp $f directdemod_amd/libdirectdemod_hip.so
  echo "== $f"
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -1
  DD_STAMPS=300 python bench.py --no-cpu-baseline --no-side --steps 3 --warmup 1 2>&1 | grep -i "stamps" | sed -n "${STAMP_LINES:-2,2p;4,5p;8,9p;12,13p;16,16p;19,19p}"
  for i in 1 2; do python bench.py --no-cpu-baseline --no-side | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
