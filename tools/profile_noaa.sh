#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf gpurun_out/prof_noaa
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_noaa -o noaa -- python3 tools/bench_noaa.py ${1:-16} > gpurun_out/prof_noaa.log 2>&1
tail -2 gpurun_out/prof_noaa.log
S=$(find gpurun_out/prof_noaa -name '*kernel_stats.csv' | head -1)
python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:28]:
    print("%-70s calls %5s  total %8.2f ms  avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
