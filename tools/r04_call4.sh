#!/bin/bash
# round 4, GPU call 4: the whole GPU suite, PMC traffic of the bench's kernel (profiles/hbm_traffic.json), kernel trace of the bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2000 python3 -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r04_call4_tests.txt
cat gpurun_out/r04_call4_tests.txt
bash tools/pmc_traffic.sh
bash tools/profile_bench.sh > gpurun_out/r04_profile_bench.txt 2>&1; tail -20 gpurun_out/r04_profile_bench.txt
