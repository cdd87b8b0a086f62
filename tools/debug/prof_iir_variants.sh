# per-kernel times of the IIR passes (tools/debug/iir_iq_time.py under rocprofv3): product library and every build/variants/lib_*.so, same call
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for l in "" $R/build/variants/lib_*.so; do
  n=$(basename "${l:-product}" .so); rm -rf $R/gpurun_out/prof_iir_$n
  echo "== $n"; DD_LIB_PATH=$l rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_iir_$n -o iir -- python3 $R/tools/debug/iir_iq_time.py 2>&1 | grep "c64 in place"
  python3 $R/tools/debug/rocpd_stats.py $R/gpurun_out/prof_iir_$n/iir_results.db iir
done
