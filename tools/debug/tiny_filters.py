import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import dd_oracle as O
from directdemod_amd import _hip, comm, filters, demod_fm, chunker
_hip.require_gpu()
fs, L = 2400000, 50000
x = O.grid_c64(O.synth_iq_fm(L, fs, 3))
for K, M in ((1, 1), (1, 7), (2, 1), (3, 5)):
    taps = np.array([0.7, -0.2, 0.4])[:K]
    class S: length = L
    ck = chunker.chunker(S(), 7001)
    out = comm.commSignal(fs // M)
    f = filters.filter(taps, 1, storeState=True); fm = demod_fm.demod_fm()
    fo = O.FilterState(taps); last = None; idx = 0; off = 0; refs = []
    for a, b in ck.getChunks:
        s = comm.commSignal(fs, x[a:b], ck).offsetFreq(25000.0).filter(f)
        if M > 1: s.bwLim(fs // M, uniq="q")
        s.funcApply(fm.demod); out.extend(s)
        y = fo.applyOn(O.nco(x[a:b], 25000.0, fs, idx)); idx += b - a
        if M > 1:
            y, r1, off, _ = O.decimate_carry(y, fs, fs // M, off)
        r, last = O.fm_demod(y, last); refs.append(r)
    ref = np.concatenate(refs)
    g = out.signal
    d = np.abs(np.angle(np.exp(1j * (g - ref))))
    print("K=%d M=%d: len %d/%d max err %.3g median %.3g kernel %d" % (K, M, len(g), len(ref), d.max(), np.median(d), f._last_kernel()))
