cd /tmp && export TMPDIR=/tmp
for v in nb2 nb4; do
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_iir_$v
DD_LIB_PATH=$GRAFT_REPO_ROOT/build/variants/lib_$v.so rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_iir_$v -o iir -- python3 $GRAFT_REPO_ROOT/tools/debug/iir_iq_time.py > /dev/null 2>&1
done
