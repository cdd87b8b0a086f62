#!/usr/bin/env python3
"""Config 4 end to end as the FIRST call of a fresh process (bench.py extra.side starts this as a child): the recording comes from a
.npy file of raw u8 pairs; prints one JSON line with the first and the second call's crude / accurate sync times."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t_imp = time.perf_counter()
from directdemod_amd import _hip, noaa_sync, source
_hip.require_gpu()
raw = np.load(sys.argv[1])
t_ready = time.perf_counter()


stages = {}


def _timed(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            stages[label] = round(stages.get(label, 0.0) + (time.perf_counter() - t) * 1e3, 2)
    setattr(obj, name, g)


from directdemod_amd import _ops, comm      # noqa: E402
_timed(noaa_sync.noaa_sync, "audio", "audio_upload_ms")
_timed(comm, "_run_batch", "chain_launch_ms")
_timed(_ops, "crude_tail", "crude_tail_ms")
_timed(_hip, "wait_copy_warmup", "wait_for_runtime_start_ms")


def one():
    src = source.IQarray(raw, 2048000)
    obj = noaa_sync.noaa_sync(src, 30000.0)
    _hip.sync()
    t0 = time.perf_counter()
    sa, sb = obj.getCrudeSync()
    _hip.sync()
    t1 = time.perf_counter()
    acc = obj.getAccurateSync()
    _hip.sync()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(sa), len(sb), len(acc[0][0]) + len(acc[1][0])


a = one()
first_stages = dict(stages)
b = one()
print(json.dumps({"iq_samples": int(raw.shape[0]), "first_call_crude_sync_ms": round(a[0], 2), "first_call_accurate_sync_ms": round(a[1], 2),
                  "first_call_total_ms": round(a[0] + a[1], 2), "second_call_total_ms": round(b[0] + b[1], 2),
                  "syncs": [a[2], a[3]], "accurate_windows": a[4], "import_and_gpu_init_s": round(t_ready - t_imp, 2),
                  "first_call_stages": first_stages}))
