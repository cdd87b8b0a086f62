"""
Pins oracle/dd_oracle.py against golden vectors produced by the reference itself
(tools/gen_golden.py, run in the build container with the reference imported from
/root/reference).  CPU only.
"""
import os

import numpy as np
import pytest

from oracle import dd_oracle as O

TOL = 1e-9


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _relerr(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b)))


@pytest.fixture(params=[0, 1, 2])
def ops(request, golden_dir):
    g = _load(golden_dir, "ops_seed%d.npz" % request.param)
    x = O.grid_c64(O.synth_iq_noise(int(g["L"]), int(g["seed"])))
    return g, x


def test_nco(ops):
    g, x = ops
    y0 = O.nco(x, 25000.0, 2400000, 0)
    y1 = O.nco(x, 25000.0, 2400000, 19999000)
    assert y0.dtype == np.complex64
    # bit-exact: same float64 phase expression, same single rounding to complex64
    assert np.array_equal(y0, g["nco_start0"])
    assert np.array_equal(y1, g["nco_start19999000"])


@pytest.mark.parametrize("name", ["hamming255", "bh151", "remez127", "gauss51", "rollavg3"])
def test_fir_stateful_uneven_chunks(ops, name):
    g, x = ops
    cuts = g["fir_cuts"]
    taps = g["taps_" + name]
    f = O.FilterState(taps)
    y = np.concatenate([f.applyOn(x[cuts[i]:cuts[i + 1]]) for i in range(3)])
    assert _relerr(y, g["fir_" + name]) < TOL
    # history form (what the GPU kernels implement) is the same filter
    hist = np.ones(len(taps) - 1, dtype=np.complex128)
    outs = []
    for i in range(3):
        yy, hist = O.fir_history_form(taps, x[cuts[i]:cuts[i + 1]], hist)
        outs.append(yy)
    assert _relerr(np.concatenate(outs), g["fir_" + name]) < TOL


def test_window_taps_match_reference(ops):
    g, _ = ops
    assert np.allclose(O.win_hamming(255), g["taps_hamming255"], atol=1e-15)
    assert np.allclose(O.win_blackmanharris(151), g["taps_bh151"], atol=1e-15)
    assert np.allclose(O.win_gaussian(51, 5), g["taps_gauss51"], atol=1e-15)
    assert np.allclose([1.0 / 3] * 3, g["taps_rollavg3"])


def test_fir_plain_and_initout(ops):
    g, x = ops
    f = O.FilterState(O.win_hamming(255), storeState=False)
    assert _relerr(f.applyOn(x), g["fir_plain_hamming255"]) < TOL
    f = O.FilterState([0.25] * 4, initOut=[1.0, 2.0, 3.0])
    y = np.concatenate([f.applyOn(x[:100].real), f.applyOn(x[100:300].real)])
    assert _relerr(y, g["fir_initout_rollavg4"]) < TOL


def test_filtfilt(ops):
    g, x = ops
    f = O.FilterState(O.win_blackmanharris(151), zeroPhase=True)
    assert _relerr(f.applyOn(x), g["filtfilt_bh151"]) < TOL
    f = O.FilterState(O.win_hamming(101), zeroPhase=True)
    assert _relerr(f.applyOn(x.real.astype(np.float64)), g["filtfilt_hamming101_real"]) < TOL
    if "filtfilt_hamming492_real" in g.files:
        f = O.FilterState(O.win_hamming(492), zeroPhase=True)
        assert _relerr(f.applyOn(x.real.astype(np.float64)), g["filtfilt_hamming492_real"]) < TOL
    with pytest.raises(ValueError):
        O.filtfilt(O.win_hamming(492), [1.0], np.zeros(1476))


def test_iir_butter(ops):
    """F4: filters.butter through the base class (lfilter DF2T with unscaled lfilter_zi,
    plain lfilter, filtfilt) -- filters.py:232-273"""
    g, x = ops
    cuts = g["fir_cuts"]
    b, a = g["iir_b"], g["iir_a"]
    xr = x.real.astype(np.float64)
    f = O.FilterState(b, a)
    y = np.concatenate([f.applyOn(xr[cuts[i]:cuts[i + 1]]) for i in range(3)])
    assert _relerr(y, g["iir_lp_real_chunks"]) < 1e-9
    assert _relerr(O.FilterState(b, a, zeroPhase=True).applyOn(xr), g["iir_lp_filtfilt"]) < 1e-9


@pytest.mark.parametrize("fs,t,tag", [(2048000, 60000, "m34"), (10000000, 200000, "m50")])
def test_decimation_carry(ops, fs, t, tag):
    g, x = ops
    cuts = g["fir_cuts"]
    off = 0
    outs = []
    for i in range(3):
        y, rate, off, M = O.decimate_carry(x[cuts[i]:cuts[i + 1]], fs, t, off)
        outs.append(y)
    y = np.concatenate(outs)
    assert np.array_equal(y, g["decim_" + tag])
    assert rate == int(g["decim_" + tag + "_rate"])
    assert np.array_equal(y, x[::M])           # Q4: global multiples of M


def test_fm(ops):
    g, _ = ops
    cuts = g["fir_cuts"]
    y = g["fir_hamming255"]
    last = None
    outs = []
    for i in range(3):
        a, last = O.fm_demod(y[cuts[i]:cuts[i + 1]], last)
        outs.append(a)
    assert np.max(np.abs(np.concatenate(outs) - g["fm_carry"])) < 1e-12
    a, _ = O.fm_demod(y, None, store_state=False)
    assert np.max(np.abs(a - g["fm_nostate"])) < 1e-12


def test_resample_strict(ops):
    g, _ = ops
    ang = g["fm_nostate"]
    y, r = O.bwlim_strict(ang, 60235, 40960)
    assert r == 40960 and _relerr(y, g["resample_60235_40960"]) < TOL
    y, r = O.bwlim_strict(ang, 200000, 11025)
    assert r == 11025 and _relerr(y, g["resample_200000_11025"]) < TOL


def test_am_envelope(ops):
    g, _ = ops
    ang = g["fm_nostate"]
    L = len(ang) + 1
    a = O.am_demod(ang[:L - 1 if (L - 1) % 2 == 0 else L - 2])
    assert _relerr(a, g["am_env_full"]) < TOL
    b = O.am_demod(np.resize(ang, 3000)) if L >= 3000 else O.am_demod(ang[:750])
    assert _relerr(b, g["am_env_3000"]) < TOL


def test_chain_chunked_bh151_m34(ops):
    g, x = ops
    L = len(x)
    assert O.chunk_list(L, 600) == g["chain_chunks"].tolist()
    y, rate = O.audio_chain(lambda a, b: x[a:b], L, 2048000, 30000.0,
                            O.win_blackmanharris(151), 60000, chunk_size=600)
    assert rate == int(g["chain_bh151_m34_rate"]) == 60235
    assert y.shape == g["chain_bh151_m34"].shape
    assert np.max(np.abs(y - g["chain_bh151_m34"])) < 1e-9


def test_chain_c2(ops):
    g, x = ops
    y, _ = O.fm_demod(O.FilterState(O.win_hamming(255)).applyOn(O.nco(x, 25000.0, 2400000)), None)
    assert np.max(np.abs(y - g["chain_c2"])) < 1e-9


def test_chain_c3(golden_dir):
    g = _load(golden_dir, "chain_c3.npz")
    L = int(g["L"])
    x = O.grid_c64(O.synth_iq_fm(L, 1e7, int(g["seed"]), f_carrier=250e3, f_mod=1e3, dev=5.0))
    y, rate = O.audio_chain(lambda a, b: x[a:b], L, 10000000, 250000.0, g["taps_remez127"],
                            200000, audio_rate=11025, strict=True, chunk_size=8192)
    assert rate == int(g["chain_c3_rate"]) == 11025
    assert y.shape == g["chain_c3"].shape
    assert np.max(np.abs(y - g["chain_c3"])) < 1e-9


@pytest.fixture(scope="module")
def noaa(golden_dir):
    g = _load(golden_dir, "noaa_c4.npz")
    raw = O.synth_apt_iq(float(g["dur"]), 2048000, seed=1)
    audio, rate = O.audio_chain(lambda a, b: O.read_iq_u8(raw, a, b), len(raw), 2048000, 30000.0,
                                O.win_blackmanharris(151), 60000, audio_rate=40960, strict=False)
    return g, raw, audio, rate


def test_noaa_audio_am_xcorr(noaa):
    g, raw, audio, rate = noaa
    assert rate == int(g["audio_rate"]) == 60235
    n3 = 3 * rate
    a3 = audio[:n3]
    assert np.max(np.abs(a3[:20000] - g["audio_3s_head"])) < 1e-9
    assert abs(np.sum(a3) - float(g["audio_3s_sum"])) < 1e-6
    am = O.am_demod_blocks(a3)
    assert np.max(np.abs(am[:20000] - g["am_3s_head"])) < 1e-9
    assert abs(np.sum(am) - float(g["am_3s_sum"])) < 1e-6
    needle = O.sync_needle(O.NOAA_SYNCA, rate)
    assert len(needle) == 560
    xc = O.xcorr_norm(am, needle)
    assert np.max(np.abs(xc[:20000] - g["xcorr_3s_syncA_head"])) < 1e-9
    assert np.max(np.abs(xc[-2000:] - g["xcorr_3s_syncA_tail"])) < 1e-9
    xc2 = O.xcorr_norm(am[:5000], needle, exact_energy=True)
    xc3 = O.xcorr_norm(am[:5000], needle, exact_energy=False)
    assert np.max(np.abs(xc2 - xc3)) < 1e-10
    # bit-exact index picks
    assert np.array_equal(O.correlate_and_find_peaks(am, rate, O.NOAA_SYNCA), g["peaks_3s_syncA"])
    assert np.array_equal(O.correlate_and_find_peaks(am, rate, O.NOAA_SYNCB), g["peaks_3s_syncB"])


def test_noaa_crude_sync_indices(noaa):
    g, raw, audio, rate = noaa
    sa, sb, useful = O.crude_sync(audio, rate)
    assert np.array_equal(sa, g["crude_syncA"])
    assert np.array_equal(sb, g["crude_syncB"])
    assert useful == int(g["useful"]) == 1


def test_noaa_accurate_sync_indices(noaa):
    g, raw, audio, rate = noaa
    fs = 2048000
    w = int(3 * O.NOAA_T * 40 * fs)
    assert w == 59076
    done = 0
    for crude, acc, pk, ts in zip(g["crude_syncA"][1:4], g["acc_syncA"][:3], g["acc_syncA_pk"], g["acc_syncA_time"]):
        c = crude / rate * fs
        start = int(c) - w
        end = int(c) + w
        assert start >= 0 and end <= len(raw)
        idx, height, tsync = O.accurate_sync_window(O.read_iq_u8(raw, start, end), fs, 30000.0, O.NOAA_SYNCA)
        assert idx + start == acc
        assert abs(height - pk) < 1e-9
        assert abs(tsync - ts) < 1e-9
        done += 1
    assert done == 3


# ----------------------------------------------------------------------------- AFSK1200 correlators
def test_afsk_correlators_match_reference_run(golden_dir):
    """sign(binary_filter) and the bit-edge signal captured from the reference's own
    decode_afsk1200.getMsg run (tools/gen_golden.py) on the audio it fed its correlator loop."""
    g = np.load(os.path.join(golden_dir, "afsk.npz"))
    tb, spb = O.afsk_tables(int(g["bw"]))
    assert tb.shape == (4, 18) and spb == 18
    bf = O.afsk_binary_filter(g["audio"], tb)
    assert np.array_equal(np.sign(bf).astype(np.int8), g["sign"])
    assert np.all(bf[-tb.shape[1]:] == 0)
    assert np.array_equal(np.where(np.arange(spb) < spb // 2, -1, 1).astype(np.int8), g["kernel"])
    ch = O.afsk_edges(bf, spb)
    assert np.array_equal(ch, g["edge_sums"].astype(np.float64) / spb)
    # the generator in the oracle reproduces the recording the golden run used
    raw = O.synth_afsk_iq(int(g["n_bits"]), int(g["fs_iq"]), int(g["seed"]))
    assert raw.shape == (int(g["n_bits"]) * int(g["fs_iq"]) // 1200, 2)


def test_config1_afsk_front_end_from_a_wav_golden(golden_dir):
    """SURVEY 8d C1 in its stated shape: the fixture was captured from the reference's own run (tools/gen_golden.py --c1: source.IQwav on a
    2.4 MS/s 8-bit stereo IQ.wav named ..._145825000Hz_IQ.wav, decode_afsk1200.getMsg's front end, decode_afsk1200.py:67-94).  The oracle's
    restatement -- read (source.py:117-118), offsetFreq 10 kHz, blackmanHarris(151) with the ones history, bwLim [::108], demod_fm --
    reproduces the reference's FM output to 1e-9 rad."""
    g = np.load(os.path.join(golden_dir, "c1_afsk_front.npz"))
    fs, M = int(g["fs"]), int(g["fs"]) // int(g["bw"])
    raw = O.synth_afsk_iq(int(g["n_bits"]), fs, int(g["seed"]), f_carrier=float(g["offset"]))
    x = O.read_iq_u8(raw, 0, raw.shape[0])
    y = O.FilterState(O.win_blackmanharris(151)).applyOn(O.nco(x, float(g["offset"]), fs, 0))[::M]
    fm, _ = O.fm_demod(y, None)
    assert M == 108 and len(fm) == int(g["n_out"])

    def dphi(a, b):
        return np.max(np.abs(np.angle(np.exp(1j * (a - b)))))
    assert dphi(fm[:2048], g["head"]) < 1e-9 and dphi(fm[-2048:], g["tail"]) < 1e-9 and dphi(fm[::4], g["every4"]) < 1e-9
