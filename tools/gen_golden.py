#!/usr/bin/env python3
"""
Generate tests/golden/*.npz by running the REFERENCE ITSELF (imported read-only
from /root/reference) on small seeded inputs.

Runs only in the build container (the reference does not exist on the GPU box and
never travels).  Only data -- inputs' seeds and the reference's outputs -- is
written.  The inputs are regenerated in the tests from the seeds through
oracle/dd_oracle.py's synthetic generators (numpy's PCG64 stream is stable).

Compat shim (harness side, the reference is untouched):
  scipy.signal.{blackmanharris,hamming,gaussian} := scipy.signal.windows.*
  scipy.signal.remez(..., Hz=Fs)                 := scipy.signal.remez(..., fs=Fs)
  scipy.ifft                                     := scipy.fft.ifft
  numpy.int                                      := int
"""
import os
import sys

import numpy as np
import scipy
import scipy.fft
import scipy.signal
import scipy.signal.windows as _w
try:
    import scipy.misc  # noqa: F401  (decode_noaa.py:13 imports scipy.misc)
except Exception:
    pass

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")


def install_shim():
    scipy.signal.blackmanharris = _w.blackmanharris
    scipy.signal.hamming = _w.hamming
    scipy.signal.gaussian = _w.gaussian
    _remez = scipy.signal.remez

    def remez(numtaps, bands, desired, weight=None, Hz=None, **kw):
        if Hz is not None:
            kw["fs"] = Hz
        return _remez(numtaps, bands, desired, weight=weight, **kw)
    scipy.signal.remez = remez
    scipy.ifft = scipy.fft.ifft
    np.int = int


class ArraySource:
    """Duck-typed stand-in for source.IQwav: same .read() arithmetic is the
    reference's own (source.py:117-118) applied to an in-memory uint8[N,2]."""

    def __init__(self, raw, fs):
        self._d = raw
        self.sampFreq = fs
        self.length = raw.shape[0]
        self.sourceType = 0

    def read(self, a, b=None):
        if b is None:
            b = a + 1
        s = self._d[a:b, 0] + 1j * self._d[a:b, 1]
        return np.array(s).astype("complex64") - (127.5 + 1j * 127.5)


def gen_afsk(O, filters):
    # ------------------------------------------------------------------ AFSK1200 correlators (decode_afsk1200.py:58-158)
    # The correlators live inline in decode_afsk1200.getMsg.  Run the reference's getMsg on a small
    # synthetic AFSK recording and capture (a) the audio it hands to the correlator loop -- the output of
    # its butter band-pass -- and (b) the arguments/result of its single np.correlate call, i.e.
    # sign(binary_filter) and the bit-edge signal; then stop it (the frame decoder behind is host logic).
    from directdemod import decode_afsk1200 as dafsk
    fs_iq = 22050 * 40
    raw = O.synth_afsk_iq(64, fs_iq, 5)
    cap = {}

    class _Stop(Exception):
        pass
    _apply = filters.filter.applyOn

    def _butter_tap(self, x):          # butter inherits applyOn: shadow it on the subclass only
        y = _apply(self, x)
        cap["audio"] = np.array(y, dtype=np.float64)
        return y
    _corr = np.correlate

    def _correlate_tap(a, v, mode="valid"):
        r = _corr(a, v, mode=mode)
        cap["sign"], cap["kernel"], cap["changes"] = np.array(a), np.array(v), np.array(r)
        raise _Stop()
    filters.butter.applyOn = _butter_tap
    dafsk.np.correlate = _correlate_tap
    try:
        dafsk.decode_afsk1200(ArraySource(raw, fs_iq), 0, 22050).getMsg
    except _Stop:
        pass
    finally:
        del filters.butter.applyOn
        dafsk.np.correlate = _corr
    spb = 22050 // 1200
    g = {"fs_iq": np.int64(fs_iq), "n_bits": np.int64(64), "seed": np.int64(5), "bw": np.int64(22050),
         "audio": cap["audio"], "sign": cap["sign"].astype(np.int8), "kernel": cap["kernel"].astype(np.int8),
         "edge_sums": np.round(cap["changes"]).astype(np.int16)}     # np.correlate output; the reference divides it by spb (:156)
    assert np.array_equal(g["edge_sums"].astype(np.float64), cap["changes"])
    np.savez_compressed(os.path.join(OUT, "afsk.npz"), **g)
    print("afsk: audio", cap["audio"].shape, "sign +/0/-", int(np.sum(cap["sign"] > 0)), int(np.sum(cap["sign"] == 0)),
          int(np.sum(cap["sign"] < 0)))


def gen_c4_60s():
    """config 4 at BENCH duration ("duration 8 s (smoke) and 60 s (bench) ... Pass = identical index
    lists"): the reference's own getCrudeSync + getAccurateSync (decode_noaa.py:769-880) over a 60 s synthetic APT
    recording (seed 1); only the index lists (a few KB) are stored.  ~10 minutes of the reference's time (its accurate
    sync is ~2 s per window), so it has its own switch:  gen_golden.py --c4-60s"""
    install_shim()
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    from directdemod import decode_noaa
    from oracle import dd_oracle as O
    dur = 60.0
    raw = O.synth_apt_iq(dur, 2048000, seed=1)
    src = ArraySource(raw, 2048000)
    nobj = decode_noaa.decode_noaa(src, 30000.0)
    sa, sb = nobj.getCrudeSync()
    g = {"dur": np.float64(dur), "seed": np.int64(1), "useful": np.int64(nobj.useful),
         "crude_syncA": np.asarray(sa, dtype=np.int64), "crude_syncB": np.asarray(sb, dtype=np.int64)}
    acc = nobj.getAccurateSync()
    g["acc_syncA"] = np.asarray(acc[0], dtype=np.int64)
    g["acc_syncB"] = np.asarray(acc[4], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "noaa_c4_60s.npz"), **g)
    print("noaa_c4_60s: crude A %d B %d, accurate A %d B %d, useful %d" % (len(sa), len(sb), len(acc[0]), len(acc[4]), nobj.useful))


C1_NAME = "synth_20180101_120000Z_145825000Hz_IQ.wav"       # (the name pattern lets main.py:167-173 parse the centre frequency)
C1_FREQ = 145835000                                         # what a user would pass as -f: the signal sits 10 kHz above the centre


def gen_c1():
    """config 1 in its stated shape (VERDICT r5 "what's missing" 3): a synthetic 8-bit stereo IQ.wav at 2.4 MS/s whose NAME
    carries the centre frequency, read by the reference's own source.IQwav, through the reference's AFSK front end exactly as getMsg runs it
    (decode_afsk1200.py:67-94: chunker -> offsetFreq -> blackmanHarris(151) -> bwLim(22050) [M = 108] -> extend; demod_fm over the whole).
    Captured: what getMsg hands to its Butterworth band-pass, i.e. the FM output.  Stored: seed and shape, the first and last 2048 angles,
    every 4th angle in between (float64) -- ~110 KB.   gen_golden.py --c1"""
    install_shim()
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tempfile
    from directdemod import decode_afsk1200 as dafsk, filters, source
    from oracle import dd_oracle as O
    from _wav import write_iq_wav
    fs, nbits, seed = 2400000, 2400, 11                      # 2 s of AFSK1200
    # the offset main.py would hand to the decoder: -f minus the "...Hz" field of the file name (main.py:167-173)
    centre = int([i for i in C1_NAME.split("_") if i[-2:] == "Hz"][0][:-2])
    offset = C1_FREQ - centre
    raw = O.synth_afsk_iq(nbits, fs, seed, f_carrier=float(offset))
    cap = {}

    class _Stop(Exception):
        pass

    def _butter_tap(self, x):
        cap["fm"] = np.array(x, dtype=np.float64)
        raise _Stop()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, C1_NAME)
        write_iq_wav(path, raw, fs)
        src = source.IQwav(path)
        assert src.sampFreq == fs and src.length == raw.shape[0]
        filters.butter.applyOn = _butter_tap
        try:
            dafsk.decode_afsk1200(src, offset, 22050).getMsg
        except _Stop:
            pass
        finally:
            del filters.butter.applyOn
    fm = cap["fm"]
    M = int(fs / 22050)
    assert M == 108 and len(fm) == len(range(0, raw.shape[0], M)) - 1
    g = {"fs": np.int64(fs), "n_bits": np.int64(nbits), "seed": np.int64(seed), "offset": np.int64(offset), "bw": np.int64(22050),
         "n_out": np.int64(len(fm)), "rate_out": np.int64(int(fs / M)), "head": fm[:2048], "tail": fm[-2048:], "every4": fm[::4]}
    np.savez_compressed(os.path.join(OUT, "c1_afsk_front.npz"), **g)
    print("c1: %d samples @%d -> %d angles @%d S/s (M = %d), offset %d Hz, fixture %d bytes" %
          (raw.shape[0], fs, len(fm), int(fs / M), M, offset, os.path.getsize(os.path.join(OUT, "c1_afsk_front.npz"))))


def main():
    if "--c4-60s" in sys.argv:
        return gen_c4_60s()
    if "--c1" in sys.argv:
        return gen_c1()
    only_afsk = "--afsk-only" in sys.argv
    install_shim()
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    import matplotlib
    matplotlib.use("Agg")
    from directdemod import comm, filters, demod_fm, demod_am, chunker, constants, decode_noaa
    from oracle import dd_oracle as O

    os.makedirs(OUT, exist_ok=True)
    if only_afsk:
        gen_afsk(O, filters)
        return

    # ------------------------------------------------------------------ per-op vectors
    for seed, L in ((0, 2048), (1, 1024), (2, 1024)):
        g = {}
        raw = O.synth_iq_noise(L, 100 + seed)
        x = O.grid_c64(raw)
        g["seed"] = np.int64(100 + seed)
        g["L"] = np.int64(L)

        # N1 offsetFreq: start index 0 (no chunker) and 19 999 000 (chunker var preset)
        s = comm.commSignal(2400000, x).offsetFreq(25000.0)
        g["nco_start0"] = s.signal.copy()

        class _Src:
            length = L
        ck = chunker.chunker(_Src())
        ck.set(constants.CHUNK_FREQOFFSET, 19999000)
        s = comm.commSignal(2400000, x, ck).offsetFreq(25000.0)
        g["nco_start19999000"] = s.signal.copy()
        assert ck.get(constants.CHUNK_FREQOFFSET) == 19999000 + L

        # F1 stateful FIR over 3 uneven chunks (one shorter than ntaps-1)
        cuts = [0, L // 2 - 37, L // 2 + 60, L]
        for name, flt in (("hamming255", filters.hamming(255)),
                          ("bh151", filters.blackmanHarris(151)),
                          ("remez127", filters.remez(10000000, [[0, 100e3], [150e3, 4999999]], [1, 0], ntaps=127)),
                          ("gauss51", filters.gaussian(51, 5)),
                          ("rollavg3", filters.rollingAverage(3))):
            outs = [flt.applyOn(x[cuts[i]:cuts[i + 1]]) for i in range(3)]
            g["fir_" + name] = np.concatenate(outs)
            g["taps_" + name] = np.asarray(flt.getB, dtype=np.float64)
        g["fir_cuts"] = np.array(cuts, dtype=np.int64)

        # F3 plain (stateless) and initOut paths
        g["fir_plain_hamming255"] = filters.hamming(255, storeState=False).applyOn(x)
        fi = filters.rollingAverage(4, initOut=[1.0, 2.0, 3.0])
        g["fir_initout_rollavg4"] = np.concatenate([fi.applyOn(x[:100].real), fi.applyOn(x[100:300].real)])

        # F2 zero-phase
        g["filtfilt_bh151"] = filters.blackmanHarris(151, zeroPhase=True).applyOn(x)
        if L > 1476:   # filtfilt needs len(x) > 3*ntaps
            g["filtfilt_hamming492_real"] = filters.hamming(492, zeroPhase=True).applyOn(x.real.astype(np.float64))
        g["filtfilt_hamming101_real"] = filters.hamming(101, zeroPhase=True).applyOn(x.real.astype(np.float64))

        # F4 butter (IIR): state carried over chunks (real + complex), plain, zero-phase
        bt = filters.butter(60235, 4160.0)
        g["iir_b"], g["iir_a"] = np.asarray(bt.getB), np.asarray(bt.getA)
        xr = x.real.astype(np.float64)
        g["iir_lp_real_chunks"] = np.concatenate([bt.applyOn(xr[cuts[i]:cuts[i + 1]]) for i in range(3)])
        btc = filters.butter(2048000, 20000.0)
        g["iir_lp_cplx_chunks"] = np.concatenate([btc.applyOn(x[cuts[i]:cuts[i + 1]]) for i in range(3)])
        g["iir_bp_plain"] = filters.butter(60235, 1000.0, 3000.0, n=4, typeFlt=constants.FLT_BP, storeState=False).applyOn(xr)
        g["iir_lp_filtfilt"] = filters.butter(60235, 4160.0, zeroPhase=True).applyOn(xr)

        # R1 decimation with carry over uneven chunks, M=34 and 50
        for fs, t, tag in ((2048000, 60000, "m34"), (10000000, 200000, "m50")):
            ck = chunker.chunker(_Src())
            outs = []
            rate = None
            for i in range(3):
                s = comm.commSignal(fs, x[cuts[i]:cuts[i + 1]], ck).bwLim(t, uniq="First")
                outs.append(np.array(s.signal))
                rate = s.sampRate
            g["decim_" + tag] = np.concatenate(outs)
            g["decim_" + tag + "_rate"] = np.int64(rate)

        # D1 FM demod with carry, on FIR output
        y = g["fir_hamming255"]
        fm = demod_fm.demod_fm()
        g["fm_carry"] = np.concatenate([fm.demod(y[cuts[i]:cuts[i + 1]]) for i in range(3)])
        g["fm_nostate"] = demod_fm.demod_fm(storeState=False).demod(y)

        # R2 strict bwLim (FFT resample) of the FM output, one chunk
        ang = g["fm_nostate"]
        s = comm.commSignal(60235, ang).bwLim(40960, True)
        g["resample_60235_40960"] = np.array(s.signal)
        s = comm.commSignal(200000, ang).bwLim(11025, True)
        g["resample_200000_11025"] = np.array(s.signal)

        # A1 AM envelope on a power-of-2 block and on a non-power-of-2 block
        am = demod_am.demod_am()
        g["am_env_full"] = am.demod(ang[:L - 1 if (L - 1) % 2 == 0 else L - 2])
        g["am_env_3000"] = am.demod(np.resize(ang, 3000)) if L >= 3000 else am.demod(ang[:750])

        # fused chain through the public API, chunked (tutorial/3_chunking.py:24-38 +
        # offsetFreq, decode_noaa.py:623), chunks of 600
        class _Src2:
            length = L
        ck = chunker.chunker(_Src2(), 600)
        out = comm.commSignal(2048000)
        bh = filters.blackmanHarris(151)
        fm = demod_fm.demod_fm()
        for a, b in ck.getChunks:
            s = comm.commSignal(2048000, x[a:b], ck).offsetFreq(30000.0).filter(bh) \
                .bwLim(60000, uniq="First").funcApply(fm.demod)
            out.extend(s)
        g["chain_bh151_m34"] = np.array(out.signal)
        g["chain_bh151_m34_rate"] = np.int64(out.sampRate)
        g["chain_chunks"] = np.array(ck.getChunks, dtype=np.int64)

        # un-decimated headline chain (C2 shape): NCO 25 kHz + hamming255 + FM, one chunk
        s = comm.commSignal(2400000, x).offsetFreq(25000.0).filter(filters.hamming(255)) \
            .funcApply(demod_fm.demod_fm().demod)
        g["chain_c2"] = np.array(s.signal)

        np.savez_compressed(os.path.join(OUT, "ops_seed%d.npz" % seed), **g)
        print("ops_seed%d: %d arrays" % (seed, len(g)))

    # ------------------------------------------------------------------ C3-shaped chain
    g = {}
    L = 40000
    raw = O.synth_iq_fm(L, 1e7, 2235, f_carrier=250e3, f_mod=1e3, dev=5.0)
    x = O.grid_c64(raw)

    class _Src3:
        length = L
    ck = chunker.chunker(_Src3(), 8192)
    out = comm.commSignal(11025)
    rz = filters.remez(10000000, [[0, 100e3], [150e3, 4999999]], [1, 0], ntaps=127)
    fm = demod_fm.demod_fm()
    for a, b in ck.getChunks:
        s = comm.commSignal(10000000, x[a:b], ck).offsetFreq(250000.0).filter(rz) \
            .bwLim(200000, uniq="First").funcApply(fm.demod).bwLim(11025, True)
        out.extend(s)
    g["seed"] = np.int64(2235)
    g["L"] = np.int64(L)
    g["chain_c3"] = np.array(out.signal)
    g["chain_c3_rate"] = np.int64(out.sampRate)
    g["taps_remez127"] = np.asarray(rz.getB, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "chain_c3.npz"), **g)
    print("chain_c3:", len(g["chain_c3"]))

    # ------------------------------------------------------------------ C4: NOAA sync
    g = {}
    dur = 6.0
    raw = O.synth_apt_iq(dur, 2048000, seed=1)
    src = ArraySource(raw, 2048000)
    nobj = decode_noaa.decode_noaa(src, 30000.0)
    sa, sb = nobj.getCrudeSync()
    g["dur"] = np.float64(dur)
    g["crude_syncA"] = np.asarray(sa, dtype=np.int64)
    g["crude_syncB"] = np.asarray(sb, dtype=np.int64)
    g["useful"] = np.int64(nobj.useful)
    acc = nobj.getAccurateSync()
    g["acc_syncA"] = np.asarray(acc[0], dtype=np.int64)
    g["acc_syncA_pk"] = np.asarray(acc[2], dtype=np.float64)
    g["acc_syncA_time"] = np.asarray([np.nan if v is None else v for v in acc[3]], dtype=np.float64)
    g["acc_syncB"] = np.asarray(acc[4], dtype=np.int64)
    g["acc_syncB_pk"] = np.asarray(acc[6], dtype=np.float64)
    g["acc_syncB_time"] = np.asarray([np.nan if v is None else v for v in acc[7]], dtype=np.float64)

    # X1/X2 on the crude-rate envelope: store the envelope-stage outputs at reduced
    # length (3 s) for the op-level test
    aud = nobj._decode_noaa__audio(constants.NOAA_CRUDESYNCSAMPRATE, False)
    g["audio_rate"] = np.int64(aud.sampRate)
    n3 = 3 * aud.sampRate
    a3 = np.array(aud.signal[:n3])
    g["audio_3s_head"] = a3.astype(np.float64)[:20000]
    g["audio_3s_sum"] = np.float64(np.sum(a3.astype(np.float64)))
    amsig = nobj._decode_noaa__getAM(comm.commSignal(aud.sampRate, a3))
    g["am_3s_head"] = np.array(amsig.signal)[:20000]
    g["am_3s_sum"] = np.float64(np.sum(np.array(amsig.signal)))
    needle = ((np.repeat(constants.NOAA_SYNCA, round(aud.sampRate * constants.NOAA_T)) * 233) + 11) / 255
    xc = np.array(nobj._decode_noaa__correlate(np.array(amsig.signal), needle))
    g["xcorr_3s_syncA_head"] = xc[:20000]
    g["xcorr_3s_syncA_tail"] = xc[-2000:]
    g["xcorr_3s_syncA_sum"] = np.float64(np.sum(xc))
    g["peaks_3s_syncA"] = np.asarray(nobj._decode_noaa__correlateAndFindPeaks(amsig, constants.NOAA_SYNCA), dtype=np.int64)
    g["peaks_3s_syncB"] = np.asarray(nobj._decode_noaa__correlateAndFindPeaks(amsig, constants.NOAA_SYNCB), dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "noaa_c4.npz"), **g)
    print("noaa_c4: crude A", g["crude_syncA"][:4], "acc A", g["acc_syncA"][:4], "useful", g["useful"])

    gen_afsk(O, filters)


if __name__ == "__main__":
    main()
