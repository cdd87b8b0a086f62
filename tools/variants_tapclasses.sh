#!/bin/bash
# tools/bench_tapclasses.py for the default library and every build/variants/lib_N.so (ablations: no parity run)
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt
for f in /tmp/lib_orig.so build/variants/lib_*.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  echo "== $f"
  timeout 300 python tools/bench_tapclasses.py 2>&1 | grep -E "${FILTER:- FM }"
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
