#!/bin/bash
# the driver's GPU tier as the driver runs it, with the per-test durations kept:  gpurun -- tools/run_gpu_tests.sh [tag]
tag=${1:-r05}
mkdir -p gpurun_out
t0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=40 > gpurun_out/${tag}_tests.txt 2>&1
rc=$?
echo "pytest rc=$rc wall=$(( $(date +%s) - t0 )) s" >> gpurun_out/${tag}_tests.txt
tail -60 gpurun_out/${tag}_tests.txt
exit $rc
