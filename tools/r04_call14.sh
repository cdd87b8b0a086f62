#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_audio.py -m gpu -q -x 2>&1 | tail -5 | cut -c1-300
python3 tools/bench_noaa.py 60 --stages 2>&1 | grep -v amdgpu.ids | tail -10
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cp build/variants/lib_1.so directdemod_amd/libdirectdemod_hip.so
echo "== arithmetic-only build (-DFF_NO_LOAD -DFF_NO_STORE -DFF_NO_LDS -DFF_FORCE_FAST), SQ passes" > gpurun_out/r04_fft1k_pmc_arith.txt
echo "time: $(KERNELS=fft1k REPS=150 ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps)" >> gpurun_out/r04_fft1k_pmc_arith.txt
KERNELS=fft1k bash tools/pmc_fft.sh >> gpurun_out/r04_fft1k_pmc_arith.txt 2>&1
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
grep -v amdgpu.ids gpurun_out/r04_fft1k_pmc_arith.txt
