#!/bin/bash
# round 4, GPU call 1: the stand-alone 8 B in / 4 B out stream sweep, WRITE_SIZE of the FFT kernel on a stream START against a
# CONTINUING chunk, and the headline kernels on SURVEY 8(d) input A (iid noise) beside input B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
./tools/ubench/bin/stream_2to1 > gpurun_out/r04_stream_2to1.txt 2>&1
echo "== input B then input A (same process each), kernels ab and fft1k" > gpurun_out/r04_inputA.txt
KERNELS=ab,fft1k REPS=150 ROUNDS=2 python3 tools/fft_ab.py >> gpurun_out/r04_inputA.txt 2>&1
INPUT=A KERNELS=ab,fft1k REPS=150 ROUNDS=2 python3 tools/fft_ab.py >> gpurun_out/r04_inputA.txt 2>&1
echo "== traffic, stream start (P.s = 1)" > gpurun_out/r04_write_size.txt
bash tools/pmc_fft_traffic.sh >> gpurun_out/r04_write_size.txt 2>&1
echo "== traffic, continuing chunk (NORESET=1: P.s = 0)" >> gpurun_out/r04_write_size.txt
NORESET=1 bash tools/pmc_fft_traffic.sh >> gpurun_out/r04_write_size.txt 2>&1
tail -5 gpurun_out/r04_stream_2to1.txt; cat gpurun_out/r04_inputA.txt gpurun_out/r04_write_size.txt
