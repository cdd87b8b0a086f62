#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -25 | cut -c1-400 > gpurun_out/r04_call11_tests.txt
cat gpurun_out/r04_call11_tests.txt
NTAPS=255 python3 tools/bench_tapclasses.py > gpurun_out/r04_tapclasses_auto.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_tapclasses_auto.txt
NTAPS=255 DD_MFMA_KERNEL=ab python3 tools/bench_tapclasses.py > gpurun_out/r04_tapclasses_ab.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_tapclasses_ab.txt
python3 tools/bench_noaa.py 60 --stages 2>&1 | grep -v amdgpu.ids | tail -9
U8=1 NTAPS=255 python3 tools/bench_tapclasses.py 2>&1 | grep -v amdgpu.ids
U8=1 NTAPS=255 DD_MFMA_KERNEL=ab python3 tools/bench_tapclasses.py 2>&1 | grep -v amdgpu.ids
