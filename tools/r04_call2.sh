#!/bin/bash
# round 4, GPU call 2: parity of the re-mapped FFT kernel, then its block -> wave map (DD_FFT_ROUNDS) x non-temporal hints x
# memory-only builds, all in one call (boxes differ by a few per cent)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_determinism.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_call2_tests.txt
cat gpurun_out/r04_call2_tests.txt
O=gpurun_out/r04_fft_map_sweep.txt
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt > $O
for f in /tmp/lib_orig.so build/variants/lib_1.so build/variants/lib_4.so build/variants/lib_5.so build/variants/lib_2.so build/variants/lib_3.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  for K in 1 2 4 8 15 29; do
    echo "== $f rounds $K: $(DD_FFT_ROUNDS=$K KERNELS=fft1k REPS=${REPS:-120} ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')" >> $O
  done
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
for K in 1 8; do
  echo "== default lib, DD_FFT_FRAME=0 (grid at output 0: misaligned stores on a stream start) rounds $K: $(DD_FFT_FRAME=0 DD_FFT_ROUNDS=$K KERNELS=fft1k REPS=${REPS:-120} ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')" >> $O
  echo "== default lib, continuing chunk (NORESET) rounds $K: $(NORESET=1 DD_FFT_ROUNDS=$K KERNELS=fft1k REPS=${REPS:-120} ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')" >> $O
  echo "== default lib, u8 input rounds $K: $(U8=1 DD_FFT_ROUNDS=$K KERNELS=fft1k REPS=${REPS:-120} ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')" >> $O
done
cat $O
