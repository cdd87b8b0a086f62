#!/bin/bash
# HBM traffic of the hot kernel from PMC counters (separate passes, MI355X_MICROARCH.md HBM/rocprofv3
# section): FETCH_SIZE and WRITE_SIZE in KiB; FETCH_SIZE doubled (gfx950 reports half the bytes of a
# wide coalesced read).  Writes gpurun_out/hbm_traffic.json + trimmed counter CSVs.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -o pmc -- python3 bench.py --steps 5 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-side --steady-ms 100 > /dev/null 2> gpurun_out/pmc_$c.err
  f=$(find gpurun_out/pmc_$c -name '*counter_collection.csv' | head -1)
  python3 tools/trim_profile.py $f gpurun_out/pmc_$c.csv
done
python3 - <<'PY'
import csv, json
import collections, os, subprocess
def mean(path, counter):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "k_chain_" in r["Kernel_Name"]:
            by[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    k = max(by, key=lambda n: len(by[n]))                   # the headline launch (most dispatches)
    return sum(by[k]) / len(by[k]), len(by[k]), k
f, nf, kname = mean("gpurun_out/pmc_FETCH_SIZE.csv", "FETCH_SIZE")
w, nw, _ = mean("gpurun_out/pmc_WRITE_SIZE.csv", "WRITE_SIZE")
try:
    git = open("gpurun_out/.git_head").read().strip()
except Exception:
    git = os.environ.get("DD_GIT_HEAD", "unknown")
rec = {"bytes_per_launch_log2n_26": int(round((2 * f + w) * 1024)), "kernel": kname.replace("void ", ""), "git": git,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 5 --warmup 1 --ramp-ms 0); "
                 "counters are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of a wide "
                 "coalesced read); the kernel named here is the one launch of a dd_chain_process call (edge rows / tiles ride along in it)",
       "FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": w, "dispatches": [nf, nw], "algorithmic_bytes": 805306368}
# one record per kernel (bench.py looks its kernel up by name); records of other kernels stay as they are
try:
    out = json.load(open("profiles/hbm_traffic.json"))
    if "kernels" not in out:
        out = {"kernels": {out["kernel"].split("<")[0]: out}}
except Exception:
    out = {"kernels": {}}
out["kernels"][rec["kernel"].split("<")[0]] = rec
json.dump(out, open("gpurun_out/hbm_traffic.json", "w"), indent=1)
print(json.dumps(rec))
PY
