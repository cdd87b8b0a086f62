#!/usr/bin/env python3
"""All-cores CPU baseline of the headline workload: one process per contiguous shard of the
sample stream, each running the reference's SciPy/NumPy calls (np.exp NCO comm.py:77, scipy.signal.lfilter with
zi filters.py:45,69, np.angle of the conj-lagged product demod_fm.py:40-49) on its shard, the shard's
(ntaps-1)-sample halo re-filtered locally.  Prints one JSON line.  Never touches the GPU; bench.py runs it as a
child process under a timeout.
usage: cpu_allcores.py [log2 samples per worker = 21] [max workers = every core of sched_getaffinity]
Workers = min(usable cores, what fits in half of MemAvailable at ~40 B/sample working set + 150 MB per
interpreter); both counts are printed."""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FS, F_OFFSET, NTAPS = 2400000, 25000.0, 255


_barrier = None


def _init(b):
    global _barrier
    _barrier = b


def work(job):
    r, lo, hi = job
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


def main():
    log2w = int(sys.argv[1]) if len(sys.argv) > 1 else 21
    usable = len(os.sched_getaffinity(0))
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else usable
    per_worker = (40 << log2w) + (150 << 20)
    fit = max(1, int(_mem_available() // 2 // per_worker))
    workers = max(1, min(cap, usable, fit))
    n = workers << log2w
    bounds = [(r, r * (1 << log2w), (r + 1) * (1 << log2w)) for r in range(workers)]
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    ctx = mp.get_context("fork")
    with ctx.Pool(workers, initializer=_init, initargs=(ctx.Barrier(workers),)) as pool:
        res = pool.map(work, bounds, chunksize=1)
    t0 = min(r[0] for r in res)
    t1 = max(r[1] for r in res)
    print(json.dumps({"value": round(n / (t1 - t0) / 1e6, 3), "unit": "MSamples/s", "cores": workers, "kind": "scipy",
                      "usable_cpus": usable, "host_cpus": os.cpu_count(),
                      "sample": "%d x 2^%d samples in contiguous shards, one process each, started together (%.2f s)"
                                % (workers, log2w, t1 - t0)}))


if __name__ == "__main__":
    main()
