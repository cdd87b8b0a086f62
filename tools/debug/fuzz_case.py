#!/usr/bin/env python3
"""one case of tests/test_gpu_audio.py::test_class_chunk_loop_random_shapes_three_ways, stage by stage"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from oracle import dd_oracle as O
from directdemod_amd import _hip, comm, filters, demod_fm, chunker, _ops
_hip.require_gpu()
rate, M, K, L, chunk, f_off, seed = 2400000, 8, 255, 347182, 160305, -30000.0, 2
raw = O.synth_iq_fm(L, rate, 50 + seed, f_carrier=f_off, f_mod=1e3, dev=5.0)
taps = O.firwin_lowpass(K, 0.4 / M)
base = _hip.DevArray.from_host(np.ascontiguousarray(raw).reshape(-1), dtype=np.uint8)
res = _hip.DevArray(L, _hip.IQ8, ptr=base.ptr, base=base)
def own(v):
    d = _hip.DevArray(v.n, v.dtype)
    _hip.check(_hip.lib().dd_memcpy_d2d(d.ptr, v.ptr, v.n * v.dtype.itemsize, None), "d2d")
    return d
def loop(get, strict):
    class S: length = L
    ck = chunker.chunker(S(), chunk)
    out = comm.commSignal(11025 if strict else rate // M)
    filt = filters.filter(taps, 1, storeState=True); fm = demod_fm.demod_fm()
    parts = []
    for a, b in ck.getChunks:
        s = comm.commSignal(rate, get(a, b), ck).offsetFreq(f_off).filter(filt).bwLim(rate // M, uniq="First").funcApply(fm.demod)
        if strict: s.bwLim(11025, True)
        out.extend(s); parts.append(s)
    sig = out.signal
    return sig, [p.signal for p in parts], filt._last_kernel()
for strict in (False, True):
    g, gp, k1 = loop(lambda a, b: res.view(a, b - a), strict)
    r, rp, k2 = loop(lambda a, b: own(res.view(a, b - a)), strict)
    print("strict", strict, "kernels", k1, k2, "equal", np.array_equal(g, r), "max diff", np.max(np.abs(g - r)))
    for i, (a, b) in enumerate(zip(gp, rp)):
        print("   chunk", i, len(a), len(b), np.array_equal(a, b), (np.max(np.abs(a - b)) if len(a) == len(b) else None))
