#!/bin/bash
# every build/variants/lib_N.so against the default library: tools/fft_ab.py with KERNELS=fft (ablation / tuning runs of k_chain_fft)
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt
for r in $(seq 1 ${ROUNDS_V:-2}); do
for f in /tmp/lib_orig.so build/variants/lib_*.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  echo "== $f: $(KERNELS=${KERNELS:-fft} REPS=${REPS:-150} ROUNDS=1 python tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')"
done
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
