#!/bin/bash
# every build/variants/lib_N.so: stamps + one steady-state bench line (ablation / tuning runs of k_chain_mfma_ab)
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt
for f in /tmp/lib_orig.so build/variants/lib_*.so; do
  [ "$f" != /tmp/lib_orig.so ] && cp $f directdemod_amd/libdirectdemod_hip.so
  echo "== $f"
  DD_STAMPS=300 python bench.py --no-cpu-baseline --no-side --steps 3 --warmup 1 2>&1 | grep -i "stamps" | sed -n "${STAMP_LINES:-2,2p;4,5p;8,9p;12,13p;16,16p;19,19p}"
  python bench.py --no-cpu-baseline --no-side | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
