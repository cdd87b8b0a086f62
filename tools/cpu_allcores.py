#!/usr/bin/env python3
"""All-cores CPU baseline of the headline workload: one process per contiguous shard of the
sample stream, each running the reference's SciPy/NumPy calls (np.exp NCO comm.py:77, scipy.signal.lfilter with
zi filters.py:45,69, np.angle of the conj-lagged product demod_fm.py:40-49) on its shard, the shard's
(ntaps-1)-sample halo re-filtered locally.  Prints one JSON line.  Never touches the GPU; bench.py runs it as a
child process under a timeout.
usage: cpu_allcores.py [log2 samples per worker = 21] [max workers = every core of sched_getaffinity]
The worker count is SWEPT (32, 64, 128, one per physical core, every hardware thread -- whichever the node has and
half of MemAvailable holds at ~40 B/sample + 150 MB per interpreter) and the best rate is reported together with the
whole sweep: these SciPy calls are memory-bound, and one worker per hardware thread is slower than fewer, pinned
workers.  Workers are pinned to distinct CPUs, one hardware thread per physical core first."""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FS, F_OFFSET, NTAPS = 2400000, 25000.0, 255


_barrier = None


_cpus = None


def _init(b, cpus):
    global _barrier, _cpus
    _barrier = b
    _cpus = cpus


def cpu_order():
    """usable CPUs, one hardware thread of every physical core first, their siblings after"""
    usable = sorted(os.sched_getaffinity(0))
    first, rest, seen = [], [], set()
    for c in usable:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except Exception:
            sib = str(c)
        if sib in seen:
            rest.append(c)
        else:
            seen.add(sib)
            first.append(c)
    return first + rest, len(first)


def work(job):
    r, lo, hi = job
    if _cpus:
        try:
            os.sched_setaffinity(0, {_cpus[r % len(_cpus)]})
        except Exception:
            pass
    import numpy as np
    import scipy.signal as ss
    from oracle import dd_oracle as O                          # input generator only
    halo = NTAPS - 1 if lo > 0 else 0
    x = O.grid_c64(O.synth_iq_fm(hi - lo + halo, FS, 1235, start=lo - halo))     # shard plus the samples before it
    b = ss.windows.hamming(NTAPS)
    _barrier.wait(timeout=120)                                # every shard's input is ready: all cores compute together
    t0 = time.time()
    sig = x
    sig *= np.exp(-1.0j * 2.0 * np.pi * F_OFFSET * np.arange(lo - halo, hi) / FS)   # absolute-index phase
    zi = ss.lfilter_zi(b, [1]) if lo == 0 else np.zeros(NTAPS - 1)               # stream start: Q1; shards: halo refills it
    y, _ = ss.lfilter(b, [1], sig, zi=zi)
    y = y[halo:] if halo else y                               # halo recomputed locally: its outputs are discarded
    a = np.angle(y[1:] * np.conj(y[:-1]))
    t1 = time.time()
    return t0, t1, len(a)


def _mem_available():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except Exception:
        pass
    return 16 << 30


def run(workers, log2w, cpus):
    n = workers << log2w
    bounds = [(r, r * (1 << log2w), (r + 1) * (1 << log2w)) for r in range(workers)]
    ctx = mp.get_context("fork")
    with ctx.Pool(workers, initializer=_init, initargs=(ctx.Barrier(workers), cpus)) as pool:
        res = pool.map(work, bounds, chunksize=1)
    t0 = min(r[0] for r in res)
    t1 = max(r[1] for r in res)
    return n / (t1 - t0) / 1e6, t1 - t0


def main():
    log2w = int(sys.argv[1]) if len(sys.argv) > 1 else 21
    cpus, physical = cpu_order()
    usable = len(cpus)
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else usable
    per_worker = (40 << log2w) + (150 << 20)
    fit = max(1, int(_mem_available() // 2 // per_worker))
    top = max(1, min(cap, usable, fit))
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    counts = sorted({w for w in (32, 64, 128, physical, top) if 1 <= w <= top})
    sweep = []
    for w in counts:
        rate, dt = run(w, log2w, cpus)
        sweep.append({"workers": w, "MSamples_per_s": round(rate, 3), "seconds": round(dt, 2)})
    best = max(sweep, key=lambda e: e["MSamples_per_s"])
    print(json.dumps({"value": best["MSamples_per_s"], "unit": "MSamples/s", "cores": best["workers"], "kind": "port", "calls": "scipy",
                      "usable_cpus": usable, "physical_cores": physical, "host_cpus": os.cpu_count(), "sweep": sweep,
                      "sample": "best of a sweep over the worker count: %d x 2^%d samples in contiguous shards, one pinned process each, "
                                "started together (%.2f s)" % (best["workers"], log2w, best["seconds"])}))


if __name__ == "__main__":
    main()
