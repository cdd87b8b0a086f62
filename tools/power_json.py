#!/usr/bin/env python3
"""profiles/power.json: shader clock and package power (rocm-smi, read-only) while each M = 1 headline kernel loops for a few seconds
(tools/debug/clock_power.py as a child process per kernel), keyed by kernel name, with the git revision.  bench.py quotes the record of
the kernel it ran as roofline.power ("not measured in this run").   usage: python tools/power_json.py [out.json]"""
import json, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "power.json")
IDLE_W, CAP_W = 283.0, 1400.0          # profiles/r03_clock_power.txt: idle at 2400 MHz; the board's cap
try:
    git = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=root).stdout.strip() or os.environ.get("DD_GIT_HEAD", "unknown")
except Exception:
    git = os.environ.get("DD_GIT_HEAD", "unknown")
if not git:
    git = os.environ.get("DD_GIT_HEAD", "unknown")
res = {"method": "rocm-smi --showclocks --showpower every 0.2 s while the kernel loops for 4 s over 2^26 resident samples (tools/debug/clock_power.py); "
                 "dynamic energy = (package power - idle) x time per launch; at the cap a launch cannot take less than dynamic energy / (cap - idle)",
       "idle_W": IDLE_W, "cap_W": CAP_W, "kernels": {}}
for sel, name, flav in (("cos1k", "k_chain_cos1k", ""), ("fft1k", "k_chain_fft1k", ""), ("ab", "k_chain_mfma_ab", ""),
                        ("cos1k", "k_chain_cos1k:u8", "u8"), ("cos1k", "k_chain_cos1k:cx", "cx")):        # (round 6: the raw-u8 and complex64-output flavours)
    env = dict(os.environ, KERNEL=sel, DUR="4", FLAVOUR=flav)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "debug", "clock_power.py")], capture_output=True, text=True, env=env)
    ms = re.search(r"= ([\d.]+) ms per launch", r.stdout)
    sc, pw = [], []
    for l in r.stdout.splitlines():
        m = re.search(r"\((\d+)Mhz\),1,\(\d+Mhz\),S,([\d.]+)", l)
        if m and "t=+" in l:
            sc.append(int(m.group(1))); pw.append(float(m.group(2)))
    sc, pw = sc[2:-1], pw[2:-1]
    if not ms or not sc:
        res["kernels"][name] = {"error": (r.stderr or r.stdout)[-300:]}
        continue
    t = float(ms.group(1)) * 1e-3
    P = sum(pw) / len(pw)
    e = (P - IDLE_W) * t
    res["kernels"][name] = {"ms_per_launch": round(t * 1e3, 4), "sclk_MHz": round(sum(sc) / len(sc)), "package_W": round(P), "samples": len(sc),
                            "dynamic_J_per_launch": round(e, 4), "ms_floor_at_the_cap": round(e / (CAP_W - IDLE_W) * 1e3, 4), "git": git}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
