"""Host-side profile of the C3 chunk loop through the drop-in classes (where do the ~150 us per chunk go?)."""
import cProfile, pstats, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from directdemod_amd import _hip, comm, filters, demod_fm, chunker
_hip.require_gpu()
fs, n, chunk = 10000000, 1 << 26, 1 << 22
rng = np.random.default_rng(2235)
x = (np.clip(np.round(60 * np.exp(2j * np.pi * 250e3 * np.arange(chunk) / fs) + 4 * (rng.standard_normal(chunk) + 1j * rng.standard_normal(chunk)) + 127.5 * (1 + 1j)), 0, 255) - 127.5 * (1 + 1j)).astype(np.complex64)
d = _hip.DevArray.from_host(np.tile(x, n // chunk))
class Src:
    length = n
    sampFreq = fs
def run():
    flt = filters.remez(fs, [[0, 100e3], [150e3, 4999999]], [1, 0], ntaps=127)
    fm = demod_fm.demod_fm()
    ck = chunker.chunker(Src(), chunk)
    audio = comm.commSignal(11025)
    _hip.sync()
    t0 = time.perf_counter()
    for a, b in ck.getChunks:
        sig = comm.commSignal(fs, d.view(a, b - a), ck).offsetFreq(250000.0).filter(flt).bwLim(200000, uniq="First") \
            .funcApply(fm.demod).bwLim(11025, True)
        audio.extend(sig)
    out = audio.device_signal
    t1 = time.perf_counter()
    _hip.sync()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t0
for _ in range(3):
    print("host issue %.3f ms, with sync %.3f ms" % tuple(1e3 * v for v in run()))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    run()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
