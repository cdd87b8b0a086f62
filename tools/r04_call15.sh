#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r04_fft_loads_first.txt
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt > $O
for r in 1 2 3; do
for f in /tmp/lib_orig.so build/variants/lib_1.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  echo "== $f: $(KERNELS=fft1k REPS=150 ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')" >> $O
done
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
echo "== default lib, u8: $(U8=1 KERNELS=fft1k REPS=150 ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')" >> $O
echo "== default lib, input A: $(INPUT=A KERNELS=fft1k REPS=150 ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')" >> $O
NTAPS=255 python3 tools/bench_tapclasses.py 2>&1 | grep -v amdgpu.ids >> $O
cat $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-300
