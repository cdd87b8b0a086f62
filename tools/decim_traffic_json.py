#!/usr/bin/env python3
"""gpurun_out/pmc_dw_<CASE>_{FETCH,WRITE}_SIZE (tools/pmc_decimw_traffic.sh) -> the "k_chain_decim_b:<CASE>" records of profiles/hbm_traffic.json,
which bench.py quotes as roofline.traffic of the decimating side entries.  Writes gpurun_out/hbm_traffic.json (copy it to profiles/)."""
import csv, glob, json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "profiles", "hbm_traffic.json")
rec = json.load(open(src)) if os.path.exists(src) else {"kernels": {}}
git = os.environ.get("GIT_REV", "")
if not git:
    try:
        git = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?"
    except Exception:
        git = "?"
n = 1 << 26
shape = {"C4": (34, 8.0), "C3": (50, 8.0), "C4u8": (34, 2.0)}
for case, (M, bin_) in shape.items():
    vals, names, cnt = {}, set(), []
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob(os.path.join(root, "gpurun_out", "pmc_dw_%s_%s" % (case, c), "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f, newline="")):
                if "k_chain_decim" in row["Kernel_Name"] and row["Counter_Name"] == c:
                    v.append(float(row["Counter_Value"]))
                    names.add(row["Kernel_Name"].split("(")[0].replace("void ", ""))
        if v:
            vals[c] = sum(v) / len(v)
            cnt.append(len(v))
    if len(vals) != 2:
        continue
    # FETCH_SIZE counts half the bytes of a 16-byte-per-lane streaming read on gfx950 (MI355X_MICROARCH.md, HBM).  The raw u8 rows are read
    # 4 bytes per lane, a width the guide calls uncalibrated: calibrated here on the known byte count -- as counted the figure is 0.517 x the
    # 2 bytes per sample the kernel must read; doubled it is 1.033 x, the complex64 rows' own ratio (same row grid, same run starts)
    wide = True
    fetch = vals["FETCH_SIZE"] * 1024 * 2
    alg = int(n * bin_ + 4 * (n // M))
    rec["kernels"]["k_chain_decim_b:" + case] = {
        "bytes_per_launch_log2n_26": int(fetch + vals["WRITE_SIZE"] * 1024), "kernel": sorted(names)[0] if names else "?", "git": git,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_decimw_traffic.sh: one 2^26-sample chunk); counters are KiB; " +
                  ("FETCH_SIZE doubled per MI355X_MICROARCH.md" + ("" if case != "C4u8" else " (4-byte-per-lane reads: the doubling calibrated on the known byte count, see tools/decim_traffic_json.py)")),
        "FETCH_SIZE_KiB_mean": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_mean": vals["WRITE_SIZE"], "dispatches": cnt, "algorithmic_bytes": alg}
    print(case, rec["kernels"]["k_chain_decim_b:" + case]["bytes_per_launch_log2n_26"], "bytes per launch,", alg, "algorithmic")
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(rec, open(os.path.join(root, "gpurun_out", "hbm_traffic.json"), "w"), indent=1)
