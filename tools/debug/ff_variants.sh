# the accurate-sync kernels of tools/bench_noaa.py 60 under rocprofv3, product library and every build/variants/lib_*.so, same call
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for l in "" $R/build/variants/lib_*.so; do
  n=$(basename "${l:-product}" .so); rm -rf $R/gpurun_out/prof_ff_$n
  DD_LIB_PATH=$l rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ff_$n -o ff -- python3 $R/tools/bench_noaa.py 60 2>&1 | grep "resident in HBM"
  echo "== $n"; python3 $R/tools/debug/rocpd_stats.py $R/gpurun_out/prof_ff_$n/ff_results.db k_filtfilt_tile
done
