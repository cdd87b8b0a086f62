#!/bin/bash
# round 4, GPU call 13: full-size suite (input A test), SQ counter passes on the arithmetic-only build of k_chain_fft1k and on the
# product kernel, the bench line with side configs and CPU baselines, kernel trace of the bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_bench_nccl.py tests/test_gpu_bench_ranks.py -m gpu -q 2>&1 | tail -8 | cut -c1-300 > gpurun_out/r04_call13_tests.txt
cat gpurun_out/r04_call13_tests.txt
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
echo "== product kernel (k_chain_fft1k<false,false>), SQ passes" > gpurun_out/r04_fft1k_pmc.txt
KERNELS=fft1k bash tools/pmc_fft.sh >> gpurun_out/r04_fft1k_pmc.txt 2>&1
cp build/variants/lib_1.so directdemod_amd/libdirectdemod_hip.so
echo "== arithmetic-only build (-DFF_NO_LOAD -DFF_NO_STORE -DFF_NO_LDS), SQ passes" >> gpurun_out/r04_fft1k_pmc.txt
echo "time: $(KERNELS=fft1k REPS=150 ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps)" >> gpurun_out/r04_fft1k_pmc.txt
KERNELS=fft1k bash tools/pmc_fft.sh >> gpurun_out/r04_fft1k_pmc.txt 2>&1
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
grep -v amdgpu.ids gpurun_out/r04_fft1k_pmc.txt
python3 bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; tail -c 6000 gpurun_out/r04_bench.json
bash tools/profile_bench.sh > gpurun_out/r04_profile_bench.txt 2>&1; head -30 gpurun_out/prof_kernel_stats.csv
