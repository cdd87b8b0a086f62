#!/bin/bash
# every benchmark of the repository in one go (headline + side measurements); output -> gpurun_out/bench_all.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
mkdir -p gpurun_out
{
  echo "== bench.py (headline, C2)";            python bench.py 2>/dev/null | tail -1
  echo "== tools/bench_configs.py";             timeout 600 python tools/bench_configs.py 2>/dev/null | tail -12
  echo "== tools/bench_u8.py";                  timeout 600 python tools/bench_u8.py 2>/dev/null | tail -6
  echo "== tools/bench_c3.py";                  timeout 600 python tools/bench_c3.py 2>/dev/null | tail -2
  echo "== tools/bench_iir.py";                 timeout 600 python tools/bench_iir.py 2>/dev/null | tail -3
  echo "== tools/bench_noaa.py 120";            timeout 600 python tools/bench_noaa.py 120 --stages 2>/dev/null | tail -8
  echo "== tools/bench_feeder.py";              timeout 600 python tools/bench_feeder.py 2>/dev/null | tail -5
  echo "== tools/ubench/bin/valu_beside_mfma";  [ -x tools/ubench/bin/valu_beside_mfma ] && ./tools/ubench/bin/valu_beside_mfma 2>/dev/null | tail -42
} > gpurun_out/bench_all.txt 2>&1
cat gpurun_out/bench_all.txt
