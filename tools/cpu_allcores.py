#!/usr/bin/env python3
"""All-cores CPU baseline of the headline workload: one process per contiguous shard of the
sample stream, each running the oracle's NumPy/SciPy restatement (NCO, 255-tap FIR with the shard's halo as
history, FM discriminator) on its shard.  Prints one JSON line.  Never touches the GPU; bench.py runs it as a
child process under a timeout.  usage: cpu_allcores.py [log2 samples per worker = 22] [max workers = 64]
(the worker count is capped so that the inputs of all shards fit comfortably in host memory: ~0.3 GB each)"""
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
    from oracle import dd_oracle as O
    halo = NTAPS - 1 if lo > 0 else 0
    x = O.grid_c64(O.synth_iq_fm(hi - lo + halo, FS, 1235, start=lo - halo))     # shard plus the samples before it
    taps = O.win_hamming(NTAPS)
    _barrier.wait(timeout=90)                                 # every shard's input is ready: all cores compute together
    t0 = time.time()
    xr = O.nco(x, F_OFFSET, FS, lo - halo)                    # absolute-index phase (no carried NCO state)
    f = O.FilterState(taps)
    y = f.applyOn(xr)[halo:] if halo else f.applyOn(xr)       # halo recomputed locally: its outputs are discarded
    a, _ = O.fm_demod(y, None)
    t1 = time.time()
    return t0, t1, len(a)


def main():
    log2w = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    workers = max(1, min(cap, len(os.sched_getaffinity(0))))
    n = workers << log2w
    bounds = [(r, r * (1 << log2w), (r + 1) * (1 << log2w)) for r in range(workers)]
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    ctx = mp.get_context("fork")
    with ctx.Pool(workers, initializer=_init, initargs=(ctx.Barrier(workers),)) as pool:
        res = pool.map(work, bounds, chunksize=1)
    t0 = min(r[0] for r in res)
    t1 = max(r[1] for r in res)
    print(json.dumps({"value": round(n / (t1 - t0) / 1e6, 3), "unit": "MSamples/s", "cores": workers,
                      "sample": "%d x 2^%d samples in contiguous shards, one process each, started together (%.2f s)"
                                % (workers, log2w, t1 - t0)}))


if __name__ == "__main__":
    main()
