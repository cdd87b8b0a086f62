#!/bin/bash
# a subset of the GPU tests with full failure text:  gpurun -- tools/run_gpu_some.sh <tag> <pytest args...>
tag=$1; shift
mkdir -p gpurun_out
python -m pytest "$@" -m gpu -q -x --durations=10 > gpurun_out/${tag}.txt 2>&1
rc=$?
tail -80 gpurun_out/${tag}.txt
exit $rc
