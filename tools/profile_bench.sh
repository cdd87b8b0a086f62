#!/bin/bash
# rocprofv3 kernel trace of the default bench run; trimmed summaries -> gpurun_out/prof_*.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof && mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof_err.log
tail -1 gpurun_out/prof_bench.json
S=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
T=$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1)
python3 tools/trim_profile.py $S gpurun_out/prof_kernel_stats.csv
cat gpurun_out/prof_kernel_stats.csv
python3 - "$T" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r["Kernel_Name"].startswith(("k_", "void k_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 12 of our kernels: name, duration, gap to previous
prev = None
for r in rows[-12:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-60s dur %8.1f us  gap %8.1f us" % (r["Kernel_Name"][:60], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
PY
