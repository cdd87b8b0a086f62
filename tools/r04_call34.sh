#!/bin/bash
# round 4, GPU call 34: decimating kernels, span (samples per tile) 3072 / 4096 / 5120 / 6144 (product) / 8192 -> 5 / 4 / 3 / 3 / 2 workgroups per CU
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04_decim_span.txt
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt > $O
for r in 1 2; do
for f in /tmp/lib_orig.so build/variants/lib_4.so build/variants/lib_1.so build/variants/lib_2.so build/variants/lib_3.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  echo "== $f: $(python3 tools/bench_configs.py 2>/dev/null | head -2 | cut -c1-120 | tr '\n' '|')" >> $O
  echo "   u8: $(python3 tools/bench_u8.py 2>/dev/null | head -1 | cut -c1-120)" >> $O
done
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
cat $O
