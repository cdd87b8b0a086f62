#!/usr/bin/env python3
"""the first audio() of a fresh process, piece by piece (each synchronised): what a cold start is made of"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
from directdemod_amd import _hip, source, constants, filters, demod_fm, comm, chunker
marks = [("imports", time.perf_counter() - t0)]
def T(label, fn):
    t = time.perf_counter(); r = fn(); _hip.sync(); marks.append((label, time.perf_counter() - t)); return r
T("require_gpu (first HIP call)", _hip.require_gpu)
raw = np.load(sys.argv[1])
d0 = T("DevArray 4 KB (first hipMalloc)", lambda: _hip.DevArray(4096, np.uint8))
src = source.IQarray(raw, 2048000)
T("read_device_raw(0, 2e7): 245 MB buffer + 40 MB upload", lambda: src.read_device_raw(0, 20000000))
T("read_device_raw(2e7, 4e7): 40 MB upload", lambda: src.read_device_raw(20000000, 40000000))
for a in range(40000000, src.length, 20000000):
    T("read_device_raw next 40 MB", lambda: src.read_device_raw(a, min(src.length, a + 20000000)))
bh = T("filters.blackmanHarris(151) (dd_fir_create, phase table)", lambda: filters.blackmanHarris(151))
fm = T("demod_fm()", lambda: demod_fm.demod_fm())
class S_: length = src.length
ck = chunker.chunker(S_(), constants.PROC_CHUNKSIZE)
out = comm.commSignal(60235)
def loop():
    for a, b in ck.getChunks:
        out.extend(comm.commSignal(src.sampFreq, src.read_device_raw(a, b), ck).offsetFreq(30000.0).filter(bh).bwLim(60000, uniq="First").funcApply(fm.demod).bwLim(60235, False))
    return out.length
T("chunk loop recorded + run (first launch of the chain kernels)", loop)
for l, v in marks:
    print("  %-64s %9.2f ms" % (l, v * 1e3))
