#!/bin/bash
# round 4, GPU call 3: the whole GPU suite on the re-mapped FFT kernel (aligned block grid, rounds, non-temporal accesses, no scratch
# copy of the kernel arguments, per-group angle path), its A/B against the plain-access build, PMC traffic, the bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_call3_tests.txt
cat gpurun_out/r04_call3_tests.txt
O=gpurun_out/r04_fft_map_sweep2.txt
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt > $O
for r in 1 2; do
for f in /tmp/lib_orig.so build/variants/lib_1.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  for K in 1 2 4 8; do
    echo "== $f rounds $K: $(DD_FFT_ROUNDS=$K KERNELS=fft1k REPS=${REPS:-150} ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')" >> $O
  done
done
done
for f in build/variants/lib_2.so build/variants/lib_3.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  for K in 1 4 8; do
    echo "== $f rounds $K: $(DD_FFT_ROUNDS=$K KERNELS=fft1k REPS=${REPS:-150} ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')" >> $O
  done
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
echo "== default lib, input A: $(INPUT=A KERNELS=fft1k,ab REPS=150 ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')" >> $O
echo "== default lib, input B: $(KERNELS=fft1k,ab REPS=150 ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')" >> $O
echo "== default lib, u8: $(U8=1 KERNELS=fft1k,ab REPS=150 ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')" >> $O
echo "== default lib, continuing chunk: $(NORESET=1 KERNELS=fft1k REPS=150 ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | tr '\n' ' ')" >> $O
cat $O
echo "== traffic, stream start (P.s = 1)" > gpurun_out/r04_write_size_after.txt
bash tools/pmc_fft_traffic.sh >> gpurun_out/r04_write_size_after.txt 2>&1
echo "== traffic, continuing chunk (NORESET=1: P.s = 0)" >> gpurun_out/r04_write_size_after.txt
NORESET=1 bash tools/pmc_fft_traffic.sh >> gpurun_out/r04_write_size_after.txt 2>&1
cat gpurun_out/r04_write_size_after.txt
python3 bench.py --no-cpu-baseline --no-side > gpurun_out/r04_bench_quick.json 2> gpurun_out/r04_bench_quick.err; cat gpurun_out/r04_bench_quick.json
