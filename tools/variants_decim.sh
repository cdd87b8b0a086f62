#!/bin/bash
# run every build/variants/lib_N.so through the decimating-chain parity tests and the front-end benchmarks
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt
for f in build/variants/lib_*.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  echo "== $f"
  [ -z "$NO_PARITY" ] && python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -1
  python tools/bench_configs.py 2>/dev/null | head -2 | cut -c1-110
  python tools/bench_u8.py | head -1
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
