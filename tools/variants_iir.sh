#!/bin/bash
# A/B of build/variants/lib_N.so on the IIR side benchmark (NO_PARITY=1 for ablation builds)
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
for f in build/variants/lib_*.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  echo "== $f"
  [ -z "$NO_PARITY" ] && python -m pytest tests/test_gpu_audio.py -m gpu -x -q -k "iir or butter" 2>&1 | tail -1
  for k in 1 2; do python tools/bench_iir.py 2>/dev/null | tail -2 | cut -c30-100 | tr "\n" "|"; echo; done
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
