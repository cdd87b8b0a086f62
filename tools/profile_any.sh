#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python script: profile_any.sh <script.py> [args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf gpurun_out/prof_any
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_any -o p -- python3 "$@" > gpurun_out/prof_any.log 2>&1
tail -3 gpurun_out/prof_any.log
S=$(find gpurun_out/prof_any -name '*kernel_stats.csv' | head -1)
python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-64s calls %5s  total %9.2f ms  avg %9.1f us" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
