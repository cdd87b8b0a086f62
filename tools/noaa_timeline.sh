#!/bin/bash
# Kernel timeline of the LAST crude + accurate sync of tools/bench_noaa.py (recording resident in HBM): start, duration and the
# idle gap before every kernel -- where the end-to-end time of config 4 goes between the kernels.  usage: tools/noaa_timeline.sh [seconds]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf gpurun_out/tl_noaa
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_noaa -o noaa -- python3 tools/bench_noaa.py ${1:-60} > gpurun_out/tl_noaa.log 2>&1
tail -1 gpurun_out/tl_noaa.log
S=$(find gpurun_out/tl_noaa -name '*kernel_trace.csv' | head -1)
python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last pass = the kernels after the last gap longer than 20 ms ... bench_noaa runs three passes; take the final third by the
# last occurrence of the audio kernel
idx = [i for i, r in enumerate(rows) if "k_chain_decim" in r["Kernel_Name"]]
rows = rows[idx[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
busy = 0
print("%9s %8s %8s  %s" % ("start us", "dur us", "gap us", "kernel"))
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:90]))
    busy += e - s
    prev_end = max(prev_end, e)
print("span %.1f us, kernels busy %.1f us, %d launches" % ((prev_end - t0) / 1e3, busy / 1e3, len(rows)))
PY
