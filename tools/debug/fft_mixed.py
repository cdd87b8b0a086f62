import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("DD_MFMA_KERNEL", "fft1k")
import directdemod_amd as dd
from directdemod_amd import comm, filters, demod_fm
from oracle import dd_oracle as O
L = 90000; fs = 2400000
x = O.grid_c64(O.synth_iq_fm(L, fs, 77, f_carrier=25e3)).astype(np.complex128)
env = np.ones(L)
prof = sys.argv[1] if len(sys.argv) > 1 else "mixed_tiles"
if prof == "mixed_tiles":
    env[20000:33000] = 1e-5; env[50000:58000] = 5e4
x = (x * env).astype(np.complex64)
taps = O.win_hamming(255)
y_ref = O.FilterState(taps).applyOn(O.nco(x, 25000.0, fs, 0))
a = comm.commSignal(fs, x).offsetFreq(25000.0).filter(filters.hamming(255)).funcApply(demod_fm.demod_fm().demod).signal
a_ref, _ = O.fm_demod(y_ref, None)
prod = np.abs(y_ref[1:] * np.conj(y_ref[:-1]))
blk = 4064
loc = np.array([np.max(prod[max(0, i - blk):i + blk]) for i in range(0, len(prod), blk)]).repeat(blk)[:len(prod)]
mask = (prod >= 1e-3 * loc) & (prod > 0)
d = np.abs(np.angle(np.exp(1j * (np.asarray(a, dtype=np.float64) - a_ref))))
dm = np.where(mask, d, 0)
idx = np.argsort(dm)[-8:]
for i in idx: print(i, dm[i], "prod/loc %.3g" % (prod[i] / loc[i]), "|y| %.4g" % abs(y_ref[i + 1]), "env", env[max(0, i - 300)], env[i], env[min(L - 1, i + 300)], "blockpos", (i + 1) % 768)
