#!/bin/bash
# rocprofv3 kernel stats of one python script:  profile_cmd.sh <tag> <script.py> [args...]  -> gpurun_out/<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
rm -rf $R/gpurun_out/prof_$TAG && mkdir -p $R/gpurun_out/prof_$TAG
cd $R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o p -- python3 "$@" > gpurun_out/prof_$TAG.out 2> gpurun_out/prof_$TAG.err < /dev/null
S=$(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)
if [ -n "$S" ]; then python3 tools/trim_profile.py "$S" gpurun_out/${TAG}_kernel_stats.csv; cut -c1-220 gpurun_out/${TAG}_kernel_stats.csv | head -${LINES_OUT:-14}; else echo "no kernel_stats.csv"; tail -5 gpurun_out/prof_$TAG.err; fi
