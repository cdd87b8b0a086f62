"""
GPU parity tests (run with -m gpu on an MI355X): the HIP path, reached through the
drop-in classes and through the raw C-ABI, against

  * the golden vectors the reference itself produced (tests/golden/*.npz), and
  * the oracle (oracle/dd_oracle.py) on other seeded inputs / sizes.

Stated float32 tolerances (the reference computes in float64):
  NCO   |err| <= 1e-6 * max|x|
  FIR   |err| <= 2e-6 * max|y|
  FM    wrapped |dphi| <= 2e-5 rad wherever |y[n] conj y[n-1]| >= 0.1 * median, <= 1e-4 rad wherever >= 1e-3 * median,
        and median |dphi| <= 2e-6 rad
Index/length/rate bookkeeping is exact.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import dd_oracle as O

pytestmark = pytest.mark.gpu

NCO_TOL = 1e-6
FIR_TOL = 2e-6
FM_MAX = 1e-4        # |z| >= 1e-3 median: the angle of a nearly cancelled product amplifies the FIR error by median/|z|
FM_WELL = 2e-5       # well-conditioned outputs (|z| >= 0.1 median; SURVEY.md's own f32 bound is 1.2e-5): VERDICT r1 weak #11
FM_MED = 2e-6


@pytest.fixture(scope="module")
def dd():
    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()
    from directdemod_amd import _hip
    _hip.require_gpu()
    import directdemod_amd.comm as comm
    import directdemod_amd.filters as filters
    import directdemod_amd.demod_fm as demod_fm
    import directdemod_amd.demod_am as demod_am
    import directdemod_amd.chunker as chunker
    import directdemod_amd._ops as _ops

    class NS:
        pass
    ns = NS()
    ns.hip, ns.comm, ns.filters, ns.demod_fm, ns.demod_am, ns.chunker, ns.ops = \
        _hip, comm, filters, demod_fm, demod_am, chunker, _ops
    return ns


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def rel_err(got, ref):
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))


def fm_check(got, ref_angle, y_ref_prod_mag=None):
    got = np.asarray(got, dtype=np.float64)
    assert got.shape == ref_angle.shape, (got.shape, ref_angle.shape)
    d = np.abs(np.angle(np.exp(1j * (got - ref_angle))))
    if y_ref_prod_mag is not None:
        mask = y_ref_prod_mag >= 1e-3 * np.median(y_ref_prod_mag)
    else:
        mask = np.ones(len(d), dtype=bool)
    assert np.max(d[mask]) <= FM_MAX, "max wrapped FM error %g" % np.max(d[mask])
    if y_ref_prod_mag is not None:
        well = y_ref_prod_mag >= 0.1 * np.median(y_ref_prod_mag)
        assert np.max(d[well]) <= FM_WELL, "max wrapped FM error on well-conditioned outputs %g" % np.max(d[well])
    assert np.median(d) <= FM_MED, "median FM error %g" % np.median(d)


class _Src:
    def __init__(self, n):
        self.length = n


@pytest.fixture(params=[0, 1, 2])
def ops(request, golden_dir):
    g = _load(golden_dir, "ops_seed%d.npz" % request.param)
    x = O.grid_c64(O.synth_iq_noise(int(g["L"]), int(g["seed"])))
    return g, x


# ----------------------------------------------------------------------------- N1
def test_nco_golden(dd, ops):
    g, x = ops
    s = dd.comm.commSignal(2400000, x).offsetFreq(25000.0)
    assert rel_err(s.signal, g["nco_start0"]) < NCO_TOL
    assert s.signal.dtype == np.complex64
    ck = dd.chunker.chunker(_Src(len(x)))
    ck.set("freqoffset", 19999000)
    s = dd.comm.commSignal(2400000, x, ck).offsetFreq(25000.0)
    assert rel_err(s.signal, g["nco_start19999000"]) < NCO_TOL


@pytest.mark.parametrize("f,fs,start", [(-123456.789, 2048000, 0), (1.0e6, 2400000, 2 ** 40 + 12345),
                                       (0.0, 1000, 5), (599999.5, 2400000, 123456789)])
def test_nco_phase_exact_for_large_index(dd, f, fs, start):
    x = O.grid_c64(O.synth_iq_noise(5000, 9))
    ck = dd.chunker.chunker(_Src(5000))
    ck.set("freqoffset", start)
    got = dd.comm.commSignal(fs, x, ck).offsetFreq(f).signal
    # oracle with exact rational phase (float64 n*f/fs loses bits at n ~ 2^40)
    from fractions import Fraction
    fr = Fraction(float(f)) / Fraction(int(fs))
    ph = np.array([float((fr * (start + i)) % 1) for i in range(5000)])
    ref = (x.astype(np.complex128) * np.exp(-2j * np.pi * ph))
    assert rel_err(got, ref) < NCO_TOL


# ----------------------------------------------------------------------------- F1
@pytest.mark.parametrize("name", ["hamming255", "bh151", "remez127", "gauss51", "rollavg3"])
def test_fir_stateful_uneven_chunks_golden(dd, ops, name):
    g, x = ops
    cuts = g["fir_cuts"]
    f = dd.filters.filter(g["taps_" + name], [1])
    y = np.concatenate([f.applyOn(x[cuts[i]:cuts[i + 1]]) for i in range(3)])
    assert y.dtype == np.complex64
    assert rel_err(y, g["fir_" + name]) < FIR_TOL


def test_fir_named_classes_golden(dd, ops):
    g, x = ops
    assert rel_err(dd.filters.hamming(255).applyOn(x), g["fir_hamming255"]) < FIR_TOL
    cuts = g["fir_cuts"]
    for cls, args, key in ((dd.filters.hamming, (255,), "hamming255"),
                           (dd.filters.blackmanHarris, (151,), "bh151"),
                           (dd.filters.gaussian, (51, 5), "gauss51"),
                           (dd.filters.rollingAverage, (3,), "rollavg3")):
        f = cls(*args)
        y = np.concatenate([f.applyOn(x[cuts[i]:cuts[i + 1]]) for i in range(3)])
        assert rel_err(y, g["fir_" + key]) < FIR_TOL


def test_fir_q1_first_sample_is_ones_history(dd):
    # Experiment 3 :276-280 on the device: [1..9] -> first output 1.0 (not 0.5)
    f = dd.filters.rollingAverage(2)
    a = f.applyOn(np.arange(1, 10).astype(np.complex64))
    b = f.applyOn(np.arange(10, 15).astype(np.complex64))
    assert np.allclose(a.real, [1.0, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5, 8.5], atol=1e-6)
    assert np.allclose(b.real, [9.5, 10.5, 11.5, 12.5, 13.5], atol=1e-6)
    # float64 real path
    f = dd.filters.rollingAverage(2)
    a = f.applyOn(np.arange(1, 10).astype(np.float64))
    b = f.applyOn(np.arange(10, 15).astype(np.float64))
    assert a.dtype == np.float64
    assert np.allclose(a, [1.0, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5, 8.5], atol=1e-12)
    assert np.allclose(b, [9.5, 10.5, 11.5, 12.5, 13.5], atol=1e-12)
    # stateless: Experiment 3 :131-132
    f = dd.filters.rollingAverage(2, storeState=False)
    assert np.allclose(f.applyOn(np.arange(1, 20).astype(np.float64)), np.arange(1, 20) - 0.5, atol=1e-12)
    assert np.allclose(f.applyOn(np.arange(10, 15).astype(np.float64))[:2], [5.0, 10.5], atol=1e-12)


def test_fir_plain_and_initout_golden(dd, ops):
    g, x = ops
    y = dd.filters.hamming(255, storeState=False).applyOn(x)
    assert rel_err(y, g["fir_plain_hamming255"]) < FIR_TOL
    y2 = dd.filters.hamming(255, storeState=False).applyOn(x)      # no state leaked
    assert np.array_equal(y, y2)
    fi = dd.filters.rollingAverage(4, initOut=[1.0, 2.0, 3.0])
    y = np.concatenate([fi.applyOn(x[:100].real.astype(np.float64)), fi.applyOn(x[100:300].real.astype(np.float64))])
    assert rel_err(y, g["fir_initout_rollavg4"]) < 1e-12
    fi = dd.filters.rollingAverage(4, initOut=[1.0, 2.0, 3.0])
    y = np.concatenate([fi.applyOn(x[:100].real.astype(np.complex64)), fi.applyOn(x[100:300].real.astype(np.complex64))])
    assert rel_err(y.real, g["fir_initout_rollavg4"]) < FIR_TOL


def test_fir_initout_float64_values_real_path(dd):
    """initOut values that are not float32 numbers must reach the float64 real path unrounded
    (filters.py:47-48,66-67 -> lfiltic; Experiment 3 cell 22 is the only user).  1/3, pi, 1e-9 + 1 differ from
    their float32 roundings by ~1e-8 relative: the first ntaps-1 outputs would show it."""
    io = [1.0 / 3.0, np.pi, 1.0 + 1e-9, -2.0 / 7.0, 1e-3 / 3.0]
    taps = np.array([0.3, -0.2, 0.25, 0.11, 0.07, -0.05, 0.9])
    x = np.random.default_rng(5).standard_normal(64)
    got = dd.filters.filter(taps, [1], initOut=io).applyOn(x)
    want = O.FilterState(taps, initOut=io).applyOn(x)
    assert got.dtype == np.float64 and np.max(np.abs(got - want)) < 1e-14 * np.max(np.abs(want)) * 10


@pytest.mark.parametrize("L", [1, 2, 7, 253, 254, 255, 2047, 2048, 2049, 4095, 5000])
def test_fir_ragged_lengths_vs_oracle(dd, L):
    x = O.grid_c64(O.synth_iq_noise(L + 300, 31 + L))
    taps = O.win_hamming(255)
    f = dd.filters.hamming(255)
    fo = O.FilterState(taps)
    got = np.concatenate([f.applyOn(x[:L]), f.applyOn(x[L:])])
    ref = np.concatenate([fo.applyOn(x[:L]), fo.applyOn(x[L:])])
    assert rel_err(got, ref) < FIR_TOL


def test_filtfilt_golden(dd, ops):
    g, x = ops
    y = dd.filters.blackmanHarris(151, zeroPhase=True).applyOn(x)
    assert rel_err(y, g["filtfilt_bh151"]) < 5e-6
    y = dd.filters.hamming(101, zeroPhase=True).applyOn(x.real.astype(np.float64))
    assert rel_err(y, g["filtfilt_hamming101_real"]) < 1e-12
    if "filtfilt_hamming492_real" in g.files:
        y = dd.filters.hamming(492, zeroPhase=True).applyOn(x.real.astype(np.float64))
        assert rel_err(y, g["filtfilt_hamming492_real"]) < 1e-12
    with pytest.raises(ValueError):
        dd.filters.hamming(492, zeroPhase=True).applyOn(np.zeros(1476))


# ----------------------------------------------------------------------------- R1 / D1
@pytest.mark.parametrize("fs,t,tag", [(2048000, 60000, "m34"), (10000000, 200000, "m50")])
def test_decimation_carry_golden(dd, ops, fs, t, tag):
    g, x = ops
    cuts = g["fir_cuts"]
    ck = dd.chunker.chunker(_Src(len(x)))
    outs = []
    for i in range(3):
        s = dd.comm.commSignal(fs, x[cuts[i]:cuts[i + 1]], ck).bwLim(t, uniq="First")
        outs.append(np.array(s.signal))
        assert s.sampRate == int(g["decim_" + tag + "_rate"])
    assert np.array_equal(np.concatenate(outs), g["decim_" + tag])     # bit-exact gather


def test_fm_golden(dd, ops):
    g, _ = ops
    cuts = g["fir_cuts"]
    y = g["fir_hamming255"]
    mag = np.abs(y[1:] * np.conj(y[:-1]))
    fm = dd.demod_fm.demod_fm()
    got = np.concatenate([fm.demod(y[cuts[i]:cuts[i + 1]]) for i in range(3)])
    fm_check(got, g["fm_carry"], mag)
    fm_check(dd.demod_fm.demod_fm(storeState=False).demod(y), g["fm_nostate"], mag)


def test_fm_notebook_vectors(dd):
    # Experiment 5: values and 1-sample carry
    x = np.array([1 + 1j, 2 - 2j, 3 + 3j, 4 - 4j, 5 + 5j, 6 - 6j])
    exp = [-np.pi / 2, np.pi / 2, -np.pi / 2, np.pi / 2, -np.pi / 2]
    assert np.allclose(dd.demod_fm.demod_fm(storeState=False).demod(x), exp, atol=1e-6)
    fm = dd.demod_fm.demod_fm()
    a, b = fm.demod(x[:3]), fm.demod(x[3:])
    assert len(a) == 2 and len(b) == 3
    assert np.allclose(np.concatenate([a, b]), exp, atol=1e-6)


# ----------------------------------------------------------------------------- fused chains
def test_chain_chunked_bh151_m34_golden(dd, ops):
    g, x = ops
    L = len(x)
    ck = dd.chunker.chunker(_Src(L), 600)
    assert ck.getChunks == g["chain_chunks"].tolist()
    out = dd.comm.commSignal(2048000)
    bh = dd.filters.blackmanHarris(151)
    fm = dd.demod_fm.demod_fm()
    for a, b in ck.getChunks:
        s = dd.comm.commSignal(2048000, x[a:b], ck).offsetFreq(30000.0).filter(bh) \
            .bwLim(60000, uniq="First").funcApply(fm.demod)
        out.extend(s)
    assert out.sampRate == int(g["chain_bh151_m34_rate"])
    # magnitude of the conj-lag product from the oracle's FIR output
    yo = O.FilterState(O.win_blackmanharris(151)).applyOn(O.nco(x, 30000.0, 2048000))[::34]
    fm_check(out.signal, g["chain_bh151_m34"], np.abs(yo[1:] * np.conj(yo[:-1])))


def test_chain_c2_golden(dd, ops):
    g, x = ops
    s = dd.comm.commSignal(2400000, x).offsetFreq(25000.0).filter(dd.filters.hamming(255)) \
        .funcApply(dd.demod_fm.demod_fm().demod)
    yo = O.FilterState(O.win_hamming(255)).applyOn(O.nco(x, 25000.0, 2400000))
    fm_check(s.signal, g["chain_c2"], np.abs(yo[1:] * np.conj(yo[:-1])))


@pytest.mark.parametrize("K", [162, 200, 255, 256])
@pytest.mark.parametrize("f_off", [25000.0, -31000.0, 0.0, 700000.0])
def test_fft_kernel_block_edges_and_carried_state(dd, K, f_off, select_kernel):
    """k_chain_fft1k (the default M = 1 FM kernel for 162..256 taps; forced here): 768-output blocks, the chunk's first and
    last block are edge blocks (carried history rotated back into the un-rotated frame, y[-1] from the carried last output,
    predicated stores, new state).  Chunks of 1, 2, K-2, 767, 768, 769, 1535, 1536 ... samples with state carried, the
    angle-subtraction form of the NCO step (|theta| <= 0.25 rad), the rotation form (700 kHz at 2.4 MS/s: theta = 1.83 rad),
    a negative offset, no offset; against the float64 oracle."""
    select_kernel("fft1k")
    fs = 2400000
    cuts = np.cumsum([0, 1, 2, K - 2, 767, 768, 769, 1535, 1536, 5000, 3, 40000, 777])
    L = int(cuts[-1])
    x = O.grid_c64(O.synth_iq_fm(L, fs, 900 + K, f_carrier=f_off if f_off else 1000.0, f_mod=700.0, dev=4.0))
    taps = O.win_hamming(K)
    flt = dd.filters.hamming(K)
    fm = dd.demod_fm.demod_fm()
    ck = dd.chunker.chunker(_Src(L))
    out = dd.comm.commSignal(fs)
    fo = O.FilterState(taps)
    last, idx, refs, mags = None, 0, [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        s = dd.comm.commSignal(fs, x[a:b], ck)
        if f_off:
            s.offsetFreq(f_off)
        s.filter(flt).funcApply(fm.demod)
        out.extend(s)
        y = fo.applyOn(O.nco(x[a:b], f_off, fs, idx) if f_off else x[a:b])
        idx += b - a
        prv = last
        r, last = O.fm_demod(y, last)
        refs.append(r)
        yy = y if prv is None else np.concatenate([[prv], y])
        mags.append(np.abs(yy[1:] * np.conj(yy[:-1])))
    ref = np.concatenate(refs)
    assert out.length == len(ref) == L - 1
    assert flt._last_kernel() == dd.hip.DD_KERNEL_FFT_OS
    fm_check(out.signal, ref, np.concatenate(mags))


@pytest.mark.parametrize("window", ["hamming", "hann_like"])
@pytest.mark.parametrize("f_off", [25000.0, -31000.0, 0.0, 700000.0])
@pytest.mark.parametrize("u8", [False, True])
def test_cos_kernel_row_edges_and_carried_state(dd, window, f_off, u8, select_kernel):
    """k_chain_cos1k (round 5: the default M = 1 FM kernel for 255 taps of the form a0 + a1 cos(2 pi k / 254), filters.py:199): the FIR
    as three running sums, rows of 1024 samples, one lane per 16 consecutive samples.  Chunks of 1, 2, 253, 1023, 1024, 1025,
    2047, 2048 ... samples with the state carried (history after the NCO, last FIR output), so that the row grid -- laid by the
    alignment of `out` -- starts anywhere, edge rows (stream start, carried state, chunk end) and interior rows alternate, and a
    wave's first row follows a row it ran without stores; complex64 and raw u8 chunks; against the float64 oracle."""
    select_kernel(None)                                          # the choice by tap class must land on it
    fs = 2400000
    cuts = np.cumsum([0, 1, 2, 253, 1023, 1024, 1025, 2047, 2048, 5000, 3, 70001, 777, 4096 * 9 + 5])
    L = int(cuts[-1])
    raw = O.synth_iq_fm(L, fs, 2900, f_carrier=f_off if f_off else 1000.0, f_mod=700.0, dev=4.0)      # (the NCO brings the carrier to 0 Hz)
    x = O.grid_c64(raw)
    if window == "hamming":
        taps = O.win_hamming(255)
        flt = dd.filters.hamming(255)
    else:
        taps = 0.5 - 0.42 * np.cos(2.0 * np.pi * np.arange(255) / 254.0)
        flt = dd.filters.filter(taps, [1])
    fm = dd.demod_fm.demod_fm()
    ck = dd.chunker.chunker(_Src(L))
    out = dd.comm.commSignal(fs)
    fo = O.FilterState(taps)
    from directdemod_amd import source
    rec = source.IQarray(raw, fs) if u8 else None
    last, idx, refs, mags = None, 0, [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        src = rec.read_device_raw(int(a), int(b)) if u8 else x[a:b]
        s = dd.comm.commSignal(fs, src, ck)
        if f_off:
            s.offsetFreq(f_off)
        s.filter(flt).funcApply(fm.demod)
        out.extend(s)
        y = fo.applyOn(O.nco(x[a:b], f_off, fs, idx) if f_off else x[a:b])
        idx += b - a
        prv = last
        r, last = O.fm_demod(y, last)
        refs.append(r)
        yy = y if prv is None else np.concatenate([[prv], y])
        mags.append(np.abs(yy[1:] * np.conj(yy[:-1])))
    ref = np.concatenate(refs)
    assert out.length == len(ref) == L - 1
    got = out.signal
    assert flt._last_kernel() == dd.hip.DD_KERNEL_COS_RS
    fm_check(got, ref, np.concatenate(mags))


@pytest.mark.parametrize("window", ["hamming", "hann_like"])
@pytest.mark.parametrize("f_off", [25000.0, 0.0])
@pytest.mark.parametrize("u8", [False, True])
def test_cos_kernel_complex_output_row_edges_and_carried_state(dd, window, f_off, u8, select_kernel):
    """k_chain_cos1k's complex64-output flavour (commSignal.filter alone, comm.py:80-92: the FIR output itself, a0 put back): the same
    ragged chunk list as the FM test above -- output rows laid by the alignment of the complex64 `out` (128-byte lines), edge rows
    with predicated stores, interior rows through the LDS transposition -- against the float64 oracle, 2e-6 of the peak."""
    select_kernel(None)
    fs = 2400000
    cuts = np.cumsum([0, 1, 2, 253, 1023, 1024, 1025, 2047, 2048, 5000, 3, 70001, 777, 4096 * 9 + 5])
    L = int(cuts[-1])
    raw = O.synth_iq_fm(L, fs, 2901, f_carrier=f_off if f_off else 1000.0, f_mod=700.0, dev=4.0)
    x = O.grid_c64(raw)
    if window == "hamming":
        taps = O.win_hamming(255)
        flt = dd.filters.hamming(255)
    else:
        taps = 0.5 - 0.42 * np.cos(2.0 * np.pi * np.arange(255) / 254.0)
        flt = dd.filters.filter(taps, [1])
    ck = dd.chunker.chunker(_Src(L))
    out = dd.comm.commSignal(fs)
    fo = O.FilterState(taps)
    from directdemod_amd import source
    rec = source.IQarray(raw, fs) if u8 else None
    idx, refs = 0, []
    for a, b in zip(cuts[:-1], cuts[1:]):
        src = rec.read_device_raw(int(a), int(b)) if u8 else x[a:b]
        s = dd.comm.commSignal(fs, src, ck)
        if f_off:
            s.offsetFreq(f_off)
        s.filter(flt)
        out.extend(s)
        refs.append(fo.applyOn(O.nco(x[a:b], f_off, fs, idx) if f_off else x[a:b]))
        idx += b - a
    ref = np.concatenate(refs)
    assert out.length == len(ref) == L
    assert flt._last_kernel() == dd.hip.DD_KERNEL_COS_RS
    assert rel_err(out.signal, ref) < FIR_TOL


def test_cos_kernel_stop_band_signal(dd, select_kernel):
    """What the running-sum form costs in accuracy, stated: a signal that lies ENTIRELY in the stop band (carrier 62 kHz from
    the pass band of Hamming 255 at 2.4 MS/s).  Its running sums R, C carry the rectangular window's side lobes (-13 dB) while their
    combination y is at -50 dB, so float32 rounding relative to the sums is several times the FFT kernel's relative to the output.  Bounds for
    k_chain_cos1k on such a signal: FIR-level error unchanged in absolute terms -- FM wrapped |dphi| <= 2e-5 rad where |z| >= 0.1 median,
    median <= 2e-6 rad -- but 1e-3 rad (not 1e-4) at the deepest nulls, |z| >= 1e-3 median.  (No float32 kernel keeps 1e-4 there on a
    signal that is all leakage: k_chain_fft1k measures 1.0e-4 to 1.1e-4 on this input and is held to 3e-4; k_chain_cos1k 1.6e-4 to 4e-4.)"""
    fs, f_off = 2400000, -31000.0
    L = 120000
    raw = O.synth_iq_fm(L, fs, 2900, f_carrier=31000.0, f_mod=700.0, dev=4.0)       # the NCO moves it to +62 kHz
    x = O.grid_c64(raw)
    y = O.FilterState(O.win_hamming(255)).applyOn(O.nco(x, f_off, fs, 0))
    ref, _ = O.fm_demod(y, None)
    mag = np.abs(y[1:] * np.conj(y[:-1]))
    res = {}
    for kern in ("cos1k", "fft1k"):
        select_kernel(None if kern == "cos1k" else kern)
        flt = dd.filters.hamming(255)
        s = dd.comm.commSignal(fs, x).offsetFreq(f_off).filter(flt).funcApply(dd.demod_fm.demod_fm().demod)
        got = np.asarray(s.signal, dtype=np.float64)
        assert flt._last_kernel() == (dd.hip.DD_KERNEL_COS_RS if kern == "cos1k" else dd.hip.DD_KERNEL_FFT_OS)
        d = np.abs(np.angle(np.exp(1j * (got - ref))))
        res[kern] = (np.median(d), np.max(d[mag >= 0.1 * np.median(mag)]), np.max(d[mag >= 1e-3 * np.median(mag)]))
    assert res["cos1k"][0] <= FM_MED and res["cos1k"][1] <= FM_WELL and res["cos1k"][2] <= 1e-3, res
    assert res["fft1k"][0] <= FM_MED and res["fft1k"][1] <= FM_WELL and res["fft1k"][2] <= 3e-4, res


def test_cos_kernel_complex_output_stop_band_signal(dd, select_kernel):
    """the FIR bound of commSignal.filter(hamming(255)) ALONE (complex64 out, no demodulator) on the stop-band input of the test
    above (max|y| is 2 % of max|x| sum|b| there), the three M = 1 kernels side by side.  Stated two ways: relative to what goes in,
    max|x| sum|b| -- each kernel <= 1e-7 (measured: k_chain_cos1k 3.9e-8, k_chain_fft1k 1.4e-8, k_chain_mfma_ab 2.7e-8; the float32 direct form's
    own level is 3.1e-7 of max|y| on pass-band signals) -- and relative to max|y|: k_chain_cos1k <= 4e-6 (measured 1.8e-6: its
    sums R, C sit 37 dB above their combination), k_chain_fft1k <= 2e-6 (6.6e-7), k_chain_mfma_ab <= 3e-6 (1.2e-6).  So the complex-output flavour
    keeps about FIR_TOL even here; what the stop band costs shows in the ANGLES at the deep nulls only (test above)."""
    fs, f_off = 2400000, -31000.0
    L = 120000
    raw = O.synth_iq_fm(L, fs, 2900, f_carrier=31000.0, f_mod=700.0, dev=4.0)
    x = O.grid_c64(raw)
    taps = O.win_hamming(255)
    y = O.FilterState(taps).applyOn(O.nco(x, f_off, fs, 0))
    scale_in = np.max(np.abs(x)) * np.sum(np.abs(taps))
    res = {}
    for kern, want in ((None, dd.hip.DD_KERNEL_COS_RS), ("fft1k", dd.hip.DD_KERNEL_FFT_OS), ("ab", dd.hip.DD_KERNEL_MFMA_AB)):
        select_kernel(kern)
        flt = dd.filters.hamming(255)
        got = np.asarray(dd.comm.commSignal(fs, x).offsetFreq(f_off).filter(flt).signal)
        assert flt._last_kernel() == want, (kern, flt._last_kernel())
        e = np.max(np.abs(got[255:] - y[255:]))                 # (behind the ones history: its step is a pass-band signal)
        res[kern or "cos1k"] = (e / scale_in, e / np.max(np.abs(y[255:])))
    assert np.max(np.abs(y[255:])) < 5e-2 * scale_in, "the input is supposed to lie in the stop band"
    print("stop-band FIR error (of max|x| sum|b|, of max|y|):", {k: ("%.3g" % a, "%.3g" % b) for k, (a, b) in res.items()})
    for k, (ein, _) in res.items():
        assert ein <= 1e-7, res
    assert res["cos1k"][1] <= 4e-6 and res["fft1k"][1] <= 2e-6 and res["ab"][1] <= 3e-6, res


def test_tight_flag_keeps_the_running_sum_kernel_off(dd, select_kernel):
    """DD_CHAIN_TIGHT / filters.filter.tight (round 6): a caller who needs the transform kernel's stop-band bound asks for
    it in public -- hamming(255) without decimation then runs k_chain_fft1k (FM output: 3e-4 rad at |z| >= 1e-3 median on the stop-band
    input instead of 1e-3), through the classes and through dd_chain_create's flags; pass-band results agree with the default kernel's
    to the FM tolerance.  (Carrying the running sums' scan state in float64 does not close the gap -- the float32 model of the kernel's
    exact operation order, tools/sim/cosfir_sim.py: 4.2e-4 -> 3.7e-4 rad on this input; profiles/r06_cos1k_stopband.txt.)"""
    import ctypes as C
    hip = dd.hip
    lib = hip.lib()
    select_kernel(None)
    fs, f_off = 2400000, -31000.0
    L = 120000
    x = O.grid_c64(O.synth_iq_fm(L, fs, 2900, f_carrier=31000.0, f_mod=700.0, dev=4.0))
    y = O.FilterState(O.win_hamming(255)).applyOn(O.nco(x, f_off, fs, 0))
    ref, _ = O.fm_demod(y, None)
    mag = np.abs(y[1:] * np.conj(y[:-1]))
    flt = dd.filters.hamming(255)
    flt.tight = True
    got = np.asarray(dd.comm.commSignal(fs, x).offsetFreq(f_off).filter(flt).funcApply(dd.demod_fm.demod_fm().demod).signal, dtype=np.float64)
    assert flt._last_kernel() == hip.DD_KERNEL_FFT_OS
    d = np.abs(np.angle(np.exp(1j * (got - ref))))
    assert np.median(d) <= FM_MED and np.max(d[mag >= 1e-3 * np.median(mag)]) <= 3e-4
    # the C-ABI: the flag of dd_chain_create
    taps = np.ascontiguousarray(O.win_hamming(255))
    src = hip.DevArray.from_host(x, dtype=np.complex64)
    for fl, want in ((0, hip.DD_KERNEL_COS_RS), (hip.DD_CHAIN_TIGHT, hip.DD_KERNEL_FFT_OS)):
        h = C.c_void_p()
        hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), 255, hip.cycles_q64(f_off, fs), 1,
                                      hip.DD_CHAIN_NCO | hip.DD_CHAIN_FM | fl))
        o = hip.DevArray(L, np.float32)
        n = C.c_int64(0)
        hip.check(lib.dd_chain_process(h, src.ptr, o.ptr, L, C.byref(n), None))
        assert lib.dd_chain_last_kernel(h) == want and n.value == L - 1
        dk = np.abs(np.angle(np.exp(1j * (o.to_host()[:L - 1].astype(np.float64) - ref))))
        assert np.max(dk[mag >= 1e-3 * np.median(mag)]) <= (3e-4 if fl else 1e-3)
        lib.dd_chain_destroy(h)


def test_cos_kernel_is_not_taken_for_a_near_pure_cosine_tap_set(dd, select_kernel):
    """k_chain_cos1k forms a1 / a0 -- a 255-tap set a0 + a1 cos(2 pi k / 254) with a0 ~ 0 (a pure cosine) fits the cosine series
    but must not take the running-sum kernel (the quotient overflows, the discriminator returns NaN): such taps fall through to the
    transform kernel.  A Hann window (a0 = a1 = 0.5) still takes it."""
    select_kernel(None)
    fs = 2400000
    L = 60000
    x = O.grid_c64(O.synth_iq_fm(L, fs, 2903, f_carrier=25000.0, f_mod=700.0, dev=4.0))
    k = np.arange(255)
    for name, taps, cos in (("pure cosine", np.cos(2 * np.pi * k / 254.0), False), ("tiny a0", 1e-9 + np.cos(2 * np.pi * k / 254.0), False),
                            ("hann", 0.5 - 0.5 * np.cos(2 * np.pi * k / 254.0), True)):
        flt = dd.filters.filter(taps, [1])
        got = np.asarray(dd.comm.commSignal(fs, x).offsetFreq(25000.0).filter(flt).funcApply(dd.demod_fm.demod_fm().demod).signal, dtype=np.float64)
        assert np.all(np.isfinite(got)), name
        assert (flt._last_kernel() == dd.hip.DD_KERNEL_COS_RS) == cos, (name, flt._last_kernel())
        y = O.FilterState(taps).applyOn(O.nco(x, 25000.0, fs, 0))
        ref, _ = O.fm_demod(y, None)
        fm_check(got, ref, np.abs(y[1:] * np.conj(y[:-1])))


@pytest.mark.parametrize("K", [162, 255, 256])
@pytest.mark.parametrize("f_off", [25000.0, -700000.0, 0.0])
@pytest.mark.parametrize("u8", [False, True])
def test_fft_kernel_complex_output_block_edges_and_carried_state(dd, K, f_off, u8, select_kernel):
    """k_chain_fft1k's complex64-output flavour (commSignal.filter without a demodulator, comm.py:80-92; round 4): the NCO factor
    e^{-j theta (abs0 + p)} applied per output as block x row pair x lane phasors, 16-byte stores, the block grid laid by the
    alignment of the OUTPUT (chunks of odd lengths put every later chunk's `out` off the 64-byte grid), edge blocks, carried
    history; chunks of 1, 2, K-2, 767 ... samples, complex64 and raw u8 chunks, against the float64 oracle."""
    select_kernel("fft1k")
    fs = 2400000
    cuts = np.cumsum([0, 1, 2, K - 2, 767, 768, 769, 1535, 1536, 5000, 3, 40001, 777])
    L = int(cuts[-1])
    raw = O.synth_iq_fm(L, fs, 1900 + K, f_carrier=abs(f_off) if f_off else 1000.0, f_mod=700.0, dev=4.0)
    x = O.grid_c64(raw)
    flt = dd.filters.hamming(K)
    fo = O.FilterState(O.win_hamming(K))
    ck = dd.chunker.chunker(_Src(L))
    out = dd.comm.commSignal(fs)
    from directdemod_amd import source
    rec = source.IQarray(raw, fs) if u8 else None        # read_device_raw: views of the recording kept in HBM as raw uint8 pairs
    refs, idx = [], 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        src = rec.read_device_raw(int(a), int(b)) if u8 else x[a:b]
        s = dd.comm.commSignal(fs, src, ck)
        if f_off:
            s.offsetFreq(f_off)
        out.extend(s.filter(flt))
        refs.append(fo.applyOn(O.nco(x[a:b], f_off, fs, idx) if f_off else x[a:b]))
        idx += b - a
    ref = np.concatenate(refs)
    assert out.length == L
    got = out.signal                                            # (chunks that are views of a resident recording run when the result is read)
    assert flt._last_kernel() == dd.hip.DD_KERNEL_FFT_OS
    assert rel_err(got, ref) < FIR_TOL


@pytest.mark.parametrize("fm", [True, False])
@pytest.mark.parametrize("kern", ["fft1k", "ab"])
def test_chunks_written_back_to_back_at_any_output_alignment(dd, fm, kern, select_kernel):
    """dd_chain_process over chunks of odd lengths whose outputs go back to back into ONE buffer: every later chunk's `out` pointer
    sits at an arbitrary element offset (4-byte aligned for angles, 8-byte for complex64).  k_chain_fft1k lays its block grid by
    that alignment (DDFft1kTabs::base: whole 64-byte lines per store instruction), so every residue 0..15 of the offset is a
    different first-block geometry; the MFMA kernel takes its own edge-tile route.  Against the float64 oracle."""
    import ctypes as C
    select_kernel(kern)
    hip = dd.hip
    lib = hip.lib()
    fs = 2400000
    lens = [3001, 70001, 33, 1, 2, 50002, 7, 767, 769, 4099, 13, 20011, 5, 3, 2049, 1025, 40003]
    L = sum(lens)
    x = O.grid_c64(O.synth_iq_fm(L, fs, 77))
    d = hip.DevArray.from_host(x)
    taps = np.ascontiguousarray(O.win_hamming(255))
    h = C.c_void_p()
    hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), 255, hip.cycles_q64(25000.0, fs), 1,
                                  hip.DD_CHAIN_NCO | (hip.DD_CHAIN_FM if fm else 0)))
    out = hip.DevArray(L + 8, np.float32 if fm else np.complex64)
    esz = 4 if fm else 8
    pos, opos, got, offs = 0, 0, C.c_int64(0), set()
    for n in lens:
        offs.add(opos % 16)
        hip.check(lib.dd_chain_process(h, d.ptr + 8 * pos, out.ptr + esz * opos, n, C.byref(got), None))
        pos += n
        opos += got.value
    assert len(offs) >= 8                                   # many different alignments of `out` were exercised
    lib.dd_chain_destroy(h)
    res = out.to_host()[:opos]
    y = O.FilterState(taps).applyOn(O.nco(x, 25000.0, fs, 0))
    if fm:
        assert opos == L - 1
        a_ref, _ = O.fm_demod(y, None)
        fm_check(res, a_ref, np.abs(y[1:] * np.conj(y[:-1])))
    else:
        assert opos == L
        assert rel_err(res, y) < FIR_TOL


def test_fft_kernel_retuned_every_chunk(dd, select_kernel):
    """A caller that changes the NCO frequency from chunk to chunk (a Doppler-tracking loop; decode_funcube.py:228 goes as far
    as a per-sample offset): k_chain_fft1k carries the NCO inside its tap spectrum, so every retune is a new spectrum -- computed
    on the host and copied in stream order into the next of four table slots, without a stream synchronisation.  Nine chunks
    (more retunes than slots, one chunk repeating its predecessor's frequency), carried FIR / FM state, against the oracle."""
    select_kernel("fft1k")
    fs, n = 2400000, 20000
    freqs = [25000.0, -40000.0, 700000.0, 1234.5, 1234.5, 300000.0, 25000.0, -1100000.0, 5.0]
    L = n * len(freqs)
    x = O.grid_c64(O.synth_iq_fm(L, fs, 4711, f_carrier=3000.0, f_mod=700.0, dev=4.0))
    flt, fm = dd.filters.hamming(255), dd.demod_fm.demod_fm()
    ck = dd.chunker.chunker(_Src(L), n)
    fo = O.FilterState(O.win_hamming(255))
    out = dd.comm.commSignal(fs)
    last, refs, mags = None, [], []
    for (a, b), f in zip(ck.getChunks, freqs):
        out.extend(dd.comm.commSignal(fs, x[a:b], ck).offsetFreq(f).filter(flt).funcApply(fm.demod))
        y = fo.applyOn(O.nco(x[a:b], f, fs, a))
        prv = last
        r, last = O.fm_demod(y, last)
        refs.append(r)
        yy = y if prv is None else np.concatenate([[prv], y])
        mags.append(np.abs(yy[1:] * np.conj(yy[:-1])))
    assert flt._last_kernel() == dd.hip.DD_KERNEL_FFT_OS
    # (the chunks' output levels differ by three orders of magnitude -- a retune to 700 kHz puts the signal deep in the stop
    # band -- so "well-conditioned" is judged against the local level, the running maximum over +-1024 outputs: the FFT kernel's
    # rounding noise is that of the largest values of the 1024-sample block an output is computed in)
    from scipy.ndimage import maximum_filter1d
    ref, prod = np.concatenate(refs), np.concatenate(mags)
    got = np.asarray(out.signal, dtype=np.float64)
    assert got.shape == ref.shape
    d = np.abs(np.angle(np.exp(1j * (got - ref))))
    loc = maximum_filter1d(prod, size=2049, mode="nearest")
    assert np.max(d[prod >= 1e-3 * loc]) <= FM_MAX
    assert np.max(d[prod >= 0.1 * loc]) <= FM_WELL
    assert np.median(d) <= FM_MED


@pytest.mark.parametrize("kern", ["fft1k", "ab"])
@pytest.mark.parametrize("seed", range(8))
def test_m1_fm_chain_random_taps_and_cuts(dd, kern, seed, select_kernel):
    """both M = 1 FM kernels on seeded random cases: taps that are no window (a low-pass design with random perturbations and sign
    changes, random length), random chunk cuts (lengths 1 .. 60 000, state carried), a random NCO frequency (either sign, up to
    fs/2) or none, raw u8 or complex64 chunks; against the float64 oracle chunk by chunk"""
    select_kernel(kern)
    rng = np.random.default_rng(7000 + seed)
    fs = int(rng.choice([2400000, 2048000, 10000000]))
    K = int(rng.integers(2, 257)) if kern == "fft1k" else int(rng.integers(16, 288))
    f_off = float(rng.uniform(-0.5, 0.5) * fs) if rng.integers(0, 4) else 0.0
    u8 = bool(rng.integers(0, 2))
    taps = O.firwin_lowpass(K, float(rng.uniform(0.02, 0.4))) if K > 3 else rng.standard_normal(K)
    taps = taps * (1.0 + 0.3 * rng.standard_normal(K)) * float(rng.choice([1.0, -1.0, 37.0]))
    ncuts = int(rng.integers(3, 9))
    lens = np.concatenate([rng.integers(1, 60000, size=ncuts), [1, int(rng.integers(1, 3 * K + 3)), 768 * int(rng.integers(1, 40))]])
    rng.shuffle(lens)
    cuts = np.concatenate([[0], np.cumsum(lens)])
    L = int(cuts[-1])
    raw = O.synth_iq_fm(L, fs, 300 + seed, f_carrier=f_off if abs(f_off) > 1 else 5000.0, f_mod=900.0, dev=3.0)
    x = O.grid_c64(raw)
    flt = dd.filters.filter(taps, 1, storeState=True)
    fm = dd.demod_fm.demod_fm()
    ck = dd.chunker.chunker(_Src(L))
    out = dd.comm.commSignal(fs)
    fo = O.FilterState(taps)
    last, idx, refs, mags, kernels = None, 0, [], [], set()
    for a, b in zip(cuts[:-1], cuts[1:]):
        a, b = int(a), int(b)
        chunk = dd.hip.DevArray.from_host(np.ascontiguousarray(raw[a:b]).reshape(-1).view(dd.hip.IQ8)) if u8 else x[a:b]
        s = dd.comm.commSignal(fs, chunk, ck)
        if f_off:
            s.offsetFreq(f_off)
        s.filter(flt).funcApply(fm.demod)
        out.extend(s)
        _ = out.length
        dd.comm.flush_all()
        kernels.add(flt._last_kernel())
        y = fo.applyOn(O.nco(x[a:b], f_off, fs, idx) if f_off else x[a:b])
        idx += b - a
        prv = last
        r, last = O.fm_demod(y, last)
        refs.append(r)
        yy = y if prv is None else np.concatenate([[prv], y])
        mags.append(np.abs(yy[1:] * np.conj(yy[:-1])))
    ref = np.concatenate(refs)
    case = dict(kern=kern, seed=seed, fs=fs, K=K, f_off=f_off, u8=u8, L=L, cuts=lens.tolist())
    assert out.length == len(ref) == L - 1, case
    if kern == "fft1k":
        assert dd.hip.DD_KERNEL_FFT_OS in kernels, (case, kernels)
    fm_check(out.signal, ref, np.concatenate(mags))


def test_fused_equals_unfused_stages(dd):
    """The fused kernel and the stage-by-stage kernels are the same arithmetic."""
    L = 30000
    x = O.grid_c64(O.synth_iq_fm(L, 2.4e6, 5))
    s = dd.comm.commSignal(2400000, x).offsetFreq(25000.0).filter(dd.filters.hamming(255)) \
        .bwLim(60000).funcApply(dd.demod_fm.demod_fm().demod)
    fused = s.signal
    a = dd.comm.commSignal(2400000, x).offsetFreq(25000.0)
    _ = a.signal
    a.filter(dd.filters.hamming(255))
    _ = a.signal
    a.bwLim(60000)
    _ = a.signal
    a.funcApply(dd.demod_fm.demod_fm().demod)
    assert fused.shape == a.signal.shape
    assert np.max(np.abs(np.angle(np.exp(1j * (fused - a.signal))))) < 2e-5


@pytest.mark.parametrize("M,K,chunk", [(1, 255, 5000), (1, 31, 777), (1, 151, 9000), (1, 127, 20000), (1, 100, 6001),
                                       (1, 257, 20000), (1, 300, 7000), (2, 64, 1000), (34, 151, 4096),
                                       (50, 127, 8192), (7, 255, 300), (200, 33, 5000), (32, 100, 3000)])
def test_fused_chain_vs_oracle_shapes(dd, M, K, chunk):
    L = 20000
    fs = 1000000
    x = O.grid_c64(O.synth_iq_fm(L, fs, 40 + M, f_carrier=20e3, f_mod=500.0, dev=3.0))
    taps = O.win_hamming(K)
    ck = dd.chunker.chunker(_Src(L), chunk)
    flt = dd.filters.filter(taps, [1])
    fm = dd.demod_fm.demod_fm()
    out = dd.comm.commSignal(fs)
    for a, b in ck.getChunks:
        s = dd.comm.commSignal(fs, x[a:b], ck).offsetFreq(20000.0).filter(flt)
        if M > 1:
            s.bwLim(fs // M, uniq="First")
        s.funcApply(fm.demod)
        out.extend(s)
    ref, rate = O.audio_chain(lambda a, b: x[a:b], L, fs, 20000.0, taps, fs // M, chunk_size=chunk)
    assert out.sampRate == rate
    yo = O.FilterState(taps).applyOn(O.nco(x, 20000.0, fs))[::M]
    fm_check(out.signal, ref, np.abs(yo[1:] * np.conj(yo[:-1])))


@pytest.mark.parametrize("M,K", [(8, 2), (8, 255), (10, 15), (12, 256), (16, 33), (32, 151), (34, 151), (40, 127), (50, 127), (62, 150), (64, 64), (64, 256)])
@pytest.mark.parametrize("fm_on", [True, False])
@pytest.mark.parametrize("u8", [False, True])
def test_decimw_kernel_shapes_cuts_and_u8(dd, M, K, fm_on, u8):
    """k_chain_decim_w (round 5: even M in 8..64, up to 256 taps): one wave per block of 2048 samples of the ABSOLUTE sample grid; M = 0 mod 8
    through the padded LDS image (gaps of two samples, zero taps over them).  Ragged
    chunk cuts (1 sample, shorter than the taps, odd lengths, blocks that straddle chunks, chunks without a kept sample) through the
    chunk-by-chunk C-ABI route against the float64 oracle; the same stream in ONE call equals the chunked outputs bit for bit (a sample
    after the NCO is a pure function of its absolute index); raw u8 input gives the bits of the same samples as complex64."""
    import ctypes as C
    hip = dd.hip
    lib = hip.lib()
    fs = 2048000
    cuts = np.cumsum([0, 1, 2, K - 1, 3, 2047, 2048, 2049, 4096 + 5, 7, 30011, 1, 20000 + M])
    L = int(cuts[-1])
    raw = O.synth_iq_fm(L, fs, 300 + M + K, f_carrier=30000.0, f_mod=900.0, dev=3.0)
    x = O.grid_c64(raw)
    taps = np.ascontiguousarray(O.firwin_lowpass(K, 0.45 / M) if K > 2 else np.array([0.5, 0.5]))
    flags = hip.DD_CHAIN_NCO | (hip.DD_CHAIN_FM if fm_on else 0) | (hip.DD_CHAIN_U8_INPUT if u8 else 0)
    src = hip.DevArray.from_host(raw.reshape(-1)) if u8 else hip.DevArray.from_host(x, dtype=np.complex64)
    isz = 2 if u8 else 8
    odt = np.float32 if fm_on else np.complex64

    def run(bounds):
        h = C.c_void_p()
        hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), K, hip.cycles_q64(30000.0, fs), M, flags))
        outs = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            no = lib.dd_chain_out_count(h, int(b - a))
            o = hip.DevArray(max(1, no), odt)
            got = C.c_int64(0)
            hip.check(lib.dd_chain_process(h, src.ptr + isz * int(a), o.ptr, int(b - a), C.byref(got), None))
            assert got.value == no
            if no:
                assert lib.dd_chain_last_kernel(h) == hip.decim_wave_kernel(K, M)
            outs.append(o.to_host()[:no])
        lib.dd_chain_destroy(h)
        return np.concatenate(outs)
    got = run(cuts)
    one = run(np.array([0, L]))
    assert got.dtype == one.dtype and np.array_equal(got.view(np.uint32), one.view(np.uint32))
    y = O.FilterState(taps).applyOn(O.nco(x, 30000.0, fs))[::M]
    if fm_on:
        ref, _ = O.fm_demod(y, None)
        fm_check(got, ref, np.abs(y[1:] * np.conj(y[:-1])))
    else:
        assert rel_err(got, y) < FIR_TOL
    if u8:
        # the same samples as complex64: the same bits
        src = hip.DevArray.from_host(x, dtype=np.complex64)
        isz, flags = 8, flags & ~hip.DD_CHAIN_U8_INPUT
        assert np.array_equal(run(np.array([0, L])).view(np.uint32), one.view(np.uint32))


@pytest.mark.parametrize("M,K", [(34, 151), (50, 127), (32, 151), (16, 33)])
def test_a_non_finite_sample_reaches_its_own_outputs_only(dd, M, K):
    """where does a NaN input sample show?  The reference's lfilter limits it to the outputs whose K-tap window holds it.
    k_chain_decim_b multiplies a block's samples with zero taps beyond K - 1 (0 x NaN), so the sample reaches the ceil(K / M) outputs whose
    BLOCKS hold it -- never fewer than the reference's, at most one more -- and nothing else: no gap cell, halo or carried sum spreads it
    (the padded image's gaps are never read; M = 32 and 16 run the padded image, 16 two passes per row).  Complex64 output, one chunk."""
    import ctypes as C
    hip = dd.hip
    lib = hip.lib()
    fs, L = 2048000, 40000
    x = O.grid_c64(O.synth_iq_fm(L, fs, 777 + M, f_carrier=30000.0, f_mod=900.0, dev=3.0)).copy()
    pos = 20011
    x[pos] = np.nan + 0j
    taps = np.ascontiguousarray(O.firwin_lowpass(K, 0.45 / M))
    h = C.c_void_p()
    hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), K, hip.cycles_q64(30000.0, fs), M, hip.DD_CHAIN_NCO))
    src = hip.DevArray.from_host(x, dtype=np.complex64)
    no = lib.dd_chain_out_count(h, L)
    o = hip.DevArray(no, np.complex64)
    got = C.c_int64(0)
    hip.check(lib.dd_chain_process(h, src.ptr, o.ptr, L, C.byref(got), None))
    assert lib.dd_chain_last_kernel(h) == hip.DD_KERNEL_DECIM_BLOCKS
    lib.dd_chain_destroy(h)
    y = o.to_host()[:got.value]
    bad = np.nonzero(~np.isfinite(y))[0]
    kept = np.arange(0, L, M)
    ref_bad = np.nonzero((kept >= pos) & (kept - (K - 1) <= pos))[0]          # the reference: outputs whose window [n - K + 1, n] holds the sample
    NI = -(-K // M)
    assert len(ref_bad) <= len(bad) <= NI and set(ref_bad) <= set(bad), (list(bad), list(ref_bad))
    assert bad.min() == ref_bad.min() and bad.max() - bad.min() == len(bad) - 1 and bad.max() <= ref_bad.max() + 1


@pytest.mark.parametrize("seed", range(40))
def test_decimw_kernel_fuzz(dd, seed):
    """seeded random shapes of k_chain_decim_w: even decimation 8..64 (plain and padded LDS images), 2..256 taps, NCO on / off (a chain without
    NCO counts every chunk from zero: the decimation phase, and with it the window alignment, then changes from chunk to chunk), FM or complex64
    output, complex64 or raw u8 input, random ragged cuts, a stream that does not start at sample 0 of the NCO's count (dd_chain_seek): the
    chunked run equals the one-call run bit for bit, and both agree with the float64 oracle."""
    import ctypes as C
    rng = np.random.default_rng(7000 + seed)
    hip = dd.hip
    lib = hip.lib()
    fs = 2048000
    M = int(rng.choice(np.arange(8, 66, 2)))
    K = int(rng.integers(2, 257))
    nco, fm_on, u8 = bool(rng.integers(0, 4) > 0), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    L = int(rng.integers(20000, 120000))
    ncut = int(rng.integers(2, 9))
    cuts = np.unique(np.concatenate([[0, L], rng.integers(1, L, size=ncut)]))
    start = int(rng.choice([0, 0, 1, 2047, 2048, 123457, 5 * 10 ** 9 + 1]))
    f_off = float(rng.choice([30000.0, -25000.0, 250000.0]))
    raw = O.synth_iq_fm(L, fs, 400 + seed, f_carrier=f_off if nco else 2000.0, f_mod=900.0, dev=3.0)
    x = O.grid_c64(raw)
    taps = np.ascontiguousarray(O.firwin_lowpass(K, 0.45 / M) if K > 2 else np.array([0.5, 0.5]))
    flags = (hip.DD_CHAIN_NCO if nco else 0) | (hip.DD_CHAIN_FM if fm_on else 0) | (hip.DD_CHAIN_U8_INPUT if u8 else 0)
    src = hip.DevArray.from_host(raw.reshape(-1)) if u8 else hip.DevArray.from_host(x, dtype=np.complex64)
    isz = 2 if u8 else 8
    odt = np.float32 if fm_on else np.complex64
    case = dict(seed=seed, M=M, K=K, nco=nco, fm=fm_on, u8=u8, L=L, cuts=cuts.tolist(), start=start)

    def run(bounds):
        h = C.c_void_p()
        hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), K, hip.cycles_q64(f_off, fs), M, flags))
        if start:
            hip.check(lib.dd_chain_seek(h, start, None))
        outs = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            no = lib.dd_chain_out_count(h, int(b - a))
            o = hip.DevArray(max(1, no), odt)
            got = C.c_int64(0)
            hip.check(lib.dd_chain_process(h, src.ptr + isz * int(a), o.ptr, int(b - a), C.byref(got), None))
            assert got.value == no, case
            if no:
                assert lib.dd_chain_last_kernel(h) == hip.decim_wave_kernel(K, M), case
            outs.append(o.to_host()[:no])
        lib.dd_chain_destroy(h)
        return np.concatenate(outs)
    got, one = run(cuts), run(np.array([0, L]))
    assert np.array_equal(got.view(np.uint32), one.view(np.uint32)), case
    xin = O.nco(x, f_off, fs, start) if nco else x
    hist = None if start == 0 else np.zeros(K - 1, dtype=np.complex128)       # dd_chain_seek: ones at the stream start (Q1), zeros elsewhere
    y = O.FilterState(taps).applyOn(xin) if hist is None else O.lfilter_fir(taps, np.concatenate([hist, xin]), None)[K - 1:]
    y = y[(M - start % M) % M::M]                             # kept global indices are the multiples of M (comm.py:123-127, Q4)
    if fm_on:
        ref, _ = O.fm_demod(y, None)
        fm_check(got, ref, np.abs(y[1:] * np.conj(y[:-1])))
    else:
        assert rel_err(got, y) < FIR_TOL, case


def test_fir_complex_output_decimated_vs_oracle(dd):
    L = 50000
    x = O.grid_c64(O.synth_iq_noise(L, 77))
    taps = O.win_blackmanharris(151)
    ck = dd.chunker.chunker(_Src(L), 7001)
    flt = dd.filters.blackmanHarris(151)
    out = dd.comm.commSignal(2048000)
    fo = O.FilterState(taps)
    refs = []
    off = 0
    for a, b in ck.getChunks:
        s = dd.comm.commSignal(2048000, x[a:b], ck).filter(flt).bwLim(60000, uniq="q")
        out.extend(s)
        y, _, off, _ = O.decimate_carry(fo.applyOn(x[a:b]), 2048000, 60000, off)
        refs.append(y)
    assert rel_err(out.signal, np.concatenate(refs)) < FIR_TOL


def test_short_first_chunks(dd):
    """chunks shorter than ntaps-1, chunks without a kept sample"""
    L = 3000
    x = O.grid_c64(O.synth_iq_fm(L, 1e6, 3, f_carrier=10e3))
    taps = O.win_hamming(255)
    cuts = [0, 5, 6, 40, 41, 300, 1000, 1001, 3000]
    for M in (1, 34):
        flt = dd.filters.hamming(255)
        fm = dd.demod_fm.demod_fm()
        ck = dd.chunker.chunker(_Src(L))
        out = dd.comm.commSignal(1000000)
        fo = O.FilterState(taps)
        last = None
        off = 0
        idx = 0
        refs = []
        mags = []
        for i in range(len(cuts) - 1):
            a, b = cuts[i], cuts[i + 1]
            s = dd.comm.commSignal(1000000, x[a:b], ck).offsetFreq(10000.0).filter(flt)
            y = fo.applyOn(O.nco(x[a:b], 10000.0, 1000000, idx))
            idx += b - a
            if M > 1:
                s.bwLim(1000000 // M, uniq="First")
                y, _, off, _ = O.decimate_carry(y, 1000000, 1000000 // M, off)
            if len(y) == 0:
                continue            # the reference's demod raises IndexError on an empty chunk
            s.funcApply(fm.demod)
            out.extend(s)
            prv = last
            r, last = O.fm_demod(y, last)
            refs.append(r)
            yy = y if prv is None else np.concatenate([[prv], y])
            mags.append(np.abs(yy[1:] * np.conj(yy[:-1])))
        ref = np.concatenate(refs)
        assert out.length == len(ref)
        # the two-tier mask of fm_check: 1e-4 rad wherever the product is not vanishing, 2e-5 on well-conditioned outputs
        fm_check(out.signal, ref, np.concatenate(mags))


# ----------------------------------------------------------------------------- raw C-ABI
def test_capi_chain_handle_and_prime(dd):
    """dd_chain_* handle: chunked == one shot; a primed shard continues the stream."""
    hip = dd.hip
    lib = hip.lib()
    L = 100000
    fs = 2400000
    x = O.grid_c64(O.synth_iq_fm(L, fs, 11))
    taps = np.ascontiguousarray(O.win_hamming(255))
    cyc = hip.cycles_q64(25000.0, fs)
    tp = taps.ctypes.data_as(C.POINTER(C.c_double))
    dx = hip.DevArray.from_host(x)

    def run(M, pieces, prime_at=None):
        h = C.c_void_p()
        hip.check(lib.dd_chain_create(C.byref(h), tp, 255, cyc, M, hip.DD_CHAIN_NCO | hip.DD_CHAIN_FM))
        outs = []
        start = 0
        if prime_at is not None:
            halo = 254 + M
            hip.check(lib.dd_chain_prime(h, dx.ptr + (prime_at - halo) * 8, halo, prime_at, None))
            start = prime_at
        for a, b in pieces:
            assert a == start
            n = b - a
            no = lib.dd_chain_out_count(h, n)
            out = hip.DevArray(max(1, no), np.float32)
            got = C.c_int64(0)
            hip.check(lib.dd_chain_process(h, dx.ptr + a * 8, out.ptr, n, C.byref(got), None))
            assert got.value == no
            outs.append(out.to_host()[:no])
            start = b
        lib.dd_chain_destroy(h)
        return np.concatenate(outs)

    for M in (1, 34):
        ref, _ = O.audio_chain(lambda a, b: x[a:b], L, fs, 25000.0, taps, fs // M if M > 1 else fs)
        one = run(M, [(0, L)])
        yo = O.FilterState(taps).applyOn(O.nco(x, 25000.0, fs))[::M]
        fm_check(one, ref, np.abs(yo[1:] * np.conj(yo[:-1])))
        chunked = run(M, [(0, 33333), (33333, 33334), (33334, 90001), (90001, L)])
        assert np.max(np.abs(np.angle(np.exp(1j * (chunked - one))))) < 2e-5
        # shard starting at 50 003: equals the tail of the one-shot result
        cut = 50003
        shard = run(M, [(cut, L)], prime_at=cut)
        n_before = len(range(0, cut, M)) - 1        # outputs produced by samples [0, cut)
        assert len(shard) == len(one) - n_before
        assert np.max(np.abs(np.angle(np.exp(1j * (shard - one[n_before:]))))) < 2e-5


def test_u8_ingest(dd):
    hip = dd.hip
    lib = hip.lib()
    raw = O.synth_iq_noise(10007, 5)
    d = hip.DevArray.from_host(raw.reshape(-1))
    out = hip.DevArray(10007, np.complex64)
    hip.check(lib.dd_u8iq_to_c64(d.ptr, out.ptr, 10007, None))
    assert np.array_equal(out.to_host(), O.grid_c64(raw))          # exact
    # fused u8 ingest inside the chain kernel == chain on the converted samples
    taps = np.ascontiguousarray(O.win_blackmanharris(151))
    tp = taps.ctypes.data_as(C.POINTER(C.c_double))
    cyc = hip.cycles_q64(30000.0, 2048000)
    res = []
    for flags, src in ((hip.DD_CHAIN_U8_INPUT, d), (0, out)):
        h = C.c_void_p()
        hip.check(lib.dd_chain_create(C.byref(h), tp, 151, cyc, 34, hip.DD_CHAIN_NCO | hip.DD_CHAIN_FM | flags))
        no = lib.dd_chain_out_count(h, 10007)
        o = hip.DevArray(no, np.float32)
        hip.check(lib.dd_chain_process(h, src.ptr, o.ptr, 10007, None, None))
        res.append(o.to_host())
        lib.dd_chain_destroy(h)
    assert np.array_equal(res[0], res[1])


def test_k_shards_on_one_gpu_equal_one_shot(dd):
    """config 5 correctness with the single GPU gpurun offers: the stream cut into k
    shards, each primed from its halo and run independently, equals the one-shot run."""
    from directdemod_amd import shard
    hip = dd.hip
    total = 300000
    fs = 2400000
    x = O.grid_c64(O.synth_iq_fm(total, fs, 21))
    dx = hip.DevArray.from_host(x)
    for M, taps in ((1, O.win_hamming(255)), (34, O.win_blackmanharris(151))):
        one = shard.HipChainEngine(taps, 25000.0, fs, M)
        out1 = hip.DevArray(total, np.float32)
        n1 = shard.run_shard(one, lambda g: dx.ptr + 8 * g, 0, total, len(taps), M, out1.ptr)
        ref = out1.to_host()[:n1]
        one.close()
        for k in (2, 3, 8):
            parts = []
            for a, b in shard.shard_ranges(total, k, M):
                eng = shard.HipChainEngine(taps, 25000.0, fs, M)
                o = hip.DevArray(max(1, b - a), np.float32)
                n = shard.run_shard(eng, lambda g: dx.ptr + 8 * g, a, b, len(taps), M, o.ptr)
                assert n == shard.output_count(a, b, M, True)
                parts.append(o.to_host()[:n])
                eng.close()
            got = np.concatenate(parts)
            assert got.shape == ref.shape
            assert np.max(np.abs(np.angle(np.exp(1j * (got.astype(np.float64) - ref))))) < 2e-5


# ----------------------------------------------------------------------------- dynamic range of the MFMA path
@pytest.mark.parametrize("profile", ["tiny", "huge", "mixed_tiles", "one_spike", "halo_spike", "zeros_then_signal"])
def test_mfma_tile_scaling_paths(dd, profile, select_kernel):
    """The f16-limb tiles are used unscaled while their peak lies in [0.25, 32768) and with a
    per-tile power-of-two scale otherwise; the decision is taken per wave with two ballots and
    published through an LDS flag.  Drive interior (persistent-kernel) tiles through every
    branch: all waves below the range, all above, a few waves out of range inside a tile,
    neighbouring tiles on different branches."""
    L = 90000                                   # ~22 tiles of 4064 outputs: interior tiles on both branches
    fs = 2400000
    x = O.grid_c64(O.synth_iq_fm(L, fs, 77, f_carrier=25e3)).astype(np.complex128)
    env = np.ones(L)
    if profile == "tiny":
        env[:] = 1e-4
    elif profile == "huge":
        env[:] = 3e5
    elif profile == "mixed_tiles":
        env[20000:33000] = 1e-5                 # whole tiles below the unit range
        env[50000:58000] = 5e4                  # whole tiles above it
    elif profile == "one_spike":
        env[41234] = 1e4                        # one sample (one wave of one tile) leaves the range
    elif profile == "halo_spike":
        # a sample above the unit range that tile 5 sees only in its 256-sample halo (its span starts at
        # 5 * 4064 - 288): in k_chain_mfma_ab the halo step's range check is the matrix wave's (a ninth slot)
        env[5 * 4064 - 288 + 70] = 3e3
    elif profile == "zeros_then_signal":
        env[:30000] = 0.0
    x = (x * env).astype(np.complex64)
    taps = O.win_hamming(255)
    # FIR + NCO, complex output
    y = dd.comm.commSignal(fs, x).offsetFreq(25000.0).filter(dd.filters.hamming(255)).signal
    y_ref = O.FilterState(taps).applyOn(O.nco(x, 25000.0, fs, 0))
    # tolerance relative to the local (tile-sized) peak: a tile's error scales with its own peak
    blk = 4064
    for s0 in range(0, L, blk):
        ref = y_ref[s0:s0 + blk]
        peak = max(np.max(np.abs(y_ref[max(0, s0 - blk):s0 + 2 * blk])), 1e-30)
        assert np.max(np.abs(y[s0:s0 + blk] - ref)) <= 4 * FIR_TOL * peak, (profile, s0)
    # FM on top (angles only where the reference product is not vanishing), through both M = 1 FM kernels.  The MFMA
    # kernel's error is relative to the peak of its 4096-output tile's neighbourhood; the FFT kernel's to the peak of the
    # 1024-sample block an output is computed in (its rounding noise is that of the block's largest values: DESIGN.md
    # 4.2c), so its reference level is the running maximum over +-1024 outputs
    from scipy.ndimage import maximum_filter1d
    a_ref, _ = O.fm_demod(y_ref, None)
    prod = np.abs(y_ref[1:] * np.conj(y_ref[:-1]))
    loc_tile = np.array([np.max(prod[max(0, i - blk):i + blk]) for i in range(0, len(prod), blk)]).repeat(blk)[:len(prod)]
    loc_fft = maximum_filter1d(prod, size=2049, mode="nearest")
    for kernel, loc in (("ab", loc_tile), ("fft1k", loc_fft)):
        select_kernel(kernel)
        a = dd.comm.commSignal(fs, x).offsetFreq(25000.0).filter(dd.filters.hamming(255)) \
            .funcApply(dd.demod_fm.demod_fm().demod).signal
        mask = prod >= 1e-3 * loc
        mask &= prod > 0
        d = np.abs(np.angle(np.exp(1j * (np.asarray(a, dtype=np.float64) - a_ref))))
        assert len(a) == L - 1
        assert np.max(d[mask]) <= 5e-5, (profile, kernel, float(np.max(d[mask])))
        if profile == "zeros_then_signal":
            # digital silence: np.angle(0) = 0 in the reference (demod_fm.py:40-49).  A block / tile whose samples are all
            # exactly zero must give exactly 0 -- not 0 * rcp(0) = NaN from the small-angle arctangent, which would
            # poison every carried filter state behind the demodulator
            assert np.all(np.isfinite(np.asarray(a))), (profile, kernel)
            assert np.all(np.asarray(a)[8192:24000] == 0.0), (profile, kernel)
            assert np.all(a_ref[8192:24000] == 0.0)


def test_seek_with_lead_in_equals_primed_shard(dd):
    """dd_chain_seek + one call over [start - lead, stop) with the lead-in's outputs dropped gives the
    outputs of dd_chain_prime + dd_chain_process (the rule bench.py's ranks > 0 use: one launch)."""
    import ctypes as C
    from directdemod_amd import shard
    hip = dd.hip
    total, fs = 260000, 2400000
    x = O.grid_c64(O.synth_iq_fm(total, fs, 33))
    dx = hip.DevArray.from_host(x)
    for M, taps, lead in ((1, O.win_hamming(255), 256), (34, O.win_blackmanharris(151), 34 * 6)):
        for start in (M * 2000, M * 3571):
            stop = total
            eng = shard.HipChainEngine(taps, 25000.0, fs, M)
            o1 = hip.DevArray(stop - start, np.float32)
            n1 = shard.run_shard(eng, lambda g: dx.ptr + 8 * g, start, stop, len(taps), M, o1.ptr)
            ref = o1.to_host()[:n1]
            eng.close()
            eng = shard.HipChainEngine(taps, 25000.0, fs, M)
            hip.check(eng.lib.dd_chain_seek(eng.h, start - lead, None), "dd_chain_seek")
            o2 = hip.DevArray(stop - start + lead, np.float32)
            n2 = eng.process(dx.ptr + 8 * (start - lead), o2.ptr, stop - start + lead)
            eng.close()
            skip = lead // M - 1                       # lead-in pairs: the first chunk of a stream is one output short (Q3)
            got = o2.to_host()[skip:n2]
            assert n2 - skip == n1 and got.shape == ref.shape
            assert np.max(np.abs(np.angle(np.exp(1j * (got.astype(np.float64) - ref))))) < 2e-5


@pytest.mark.parametrize("L", [300, 4063, 4064, 4065, 4096 + 254, 8128, 8129, 12191, 12192, 12193, 16257, 40000])
@pytest.mark.parametrize("shift", [0, 1])
def test_mfma_tile_boundaries_and_alignment(dd, L, shift):
    """chunk lengths around the 4064-output tile advance (no interior tile, exactly one, ragged last
    tile) and an input that starts on an 8-byte but not 16-byte boundary (every tile then takes the
    predicated edge kernel); FM and complex output, two chunks so the carried state crosses too"""
    fs = 2400000
    x = O.grid_c64(O.synth_iq_fm(L + shift + 5000, fs, 100 + L % 97))
    d = dd.hip.DevArray.from_host(x)
    taps = O.win_hamming(255)
    for fm in (True, False):
        f = dd.filters.hamming(255)
        dem = dd.demod_fm.demod_fm()
        ck = dd.chunker.chunker(_Src(L + 5000))
        outs = []
        for a, b in ((0, L), (L, L + 5000)):
            s = dd.comm.commSignal(fs, d.view(shift + a, b - a), ck).offsetFreq(25000.0).filter(f)
            if fm:
                s = s.funcApply(dem.demod)
            outs.append(np.asarray(s.signal))
        got = np.concatenate(outs)
        xs = x[shift:shift + L + 5000]
        y_ref = O.FilterState(taps).applyOn(O.nco(xs, 25000.0, fs, 0))
        if fm:
            a_ref, _ = O.fm_demod(y_ref, None)
            fm_check(got, a_ref, np.abs(y_ref[1:] * np.conj(y_ref[:-1])))
        else:
            assert rel_err(got, y_ref) < FIR_TOL


def test_nco_per_sample_frequency_array(dd):
    """offsetFreq with an array (Doppler correction, decode_funcube.py:228), with and without a chunker offset"""
    fs, L = 2400000, 30000
    x = O.grid_c64(O.synth_iq_noise(L, 4))
    f = 25000.0 + 300.0 * np.sin(2 * np.pi * np.arange(L) / 7000.0)
    got = dd.comm.commSignal(fs, x).offsetFreq(f).signal
    assert got.dtype == np.complex64 and rel_err(got, O.nco(x, f, fs, 0)) < NCO_TOL
    ck = dd.chunker.chunker(_Src(L))
    ck.set("freqoffset", 4000000)
    got = dd.comm.commSignal(fs, x, ck).offsetFreq(f).filter(dd.filters.hamming(255)).signal
    ref = O.FilterState(O.win_hamming(255)).applyOn(O.nco(x, f, fs, 4000000))
    assert ck.get("freqoffset") == 4000000 + L
    assert rel_err(got, ref) < FIR_TOL
    with pytest.raises(ValueError):
        dd.comm.commSignal(fs, x).offsetFreq(f[:-1])


@pytest.mark.parametrize("M,fm", [(34, True), (50, True), (34, False), (3, True)])
def test_u8_ingest_long_chunks_persistent_path(dd, M, fm):
    """raw u8 input long enough for the persistent decimating kernel's u8 flavour (>= 64 interior tiles),
    two chunks with carried state, against the oracle on the widened samples"""
    import ctypes as C
    hip = dd.hip
    lib = hip.lib()
    fs = 2048000
    L1, L2 = 64 * 6144 * 2 + 12345, 300001
    raw = O.synth_iq_fm(L1 + L2, fs, 41, f_carrier=30e3)
    d = hip.DevArray.from_host(raw.reshape(-1))
    taps = np.ascontiguousarray(O.win_blackmanharris(151))
    h = C.c_void_p()
    hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), 151, hip.cycles_q64(30000.0, fs), M,
                                  hip.DD_CHAIN_NCO | hip.DD_CHAIN_U8_INPUT | (hip.DD_CHAIN_FM if fm else 0)))
    outs = []
    pos = 0
    for n in (L1, L2):
        no = lib.dd_chain_out_count(h, n)
        o = hip.DevArray(max(1, no), np.float32 if fm else np.complex64)
        got = C.c_int64(0)
        hip.check(lib.dd_chain_process(h, d.ptr + 2 * pos, o.ptr, n, C.byref(got), None))
        assert got.value == no
        outs.append(o.to_host()[:no])
        pos += n
    lib.dd_chain_destroy(h)
    got = np.concatenate(outs)
    x = O.grid_c64(raw)
    y = O.FilterState(taps).applyOn(O.nco(x, 30000.0, fs, 0))[::M]
    if fm:
        a_ref, _ = O.fm_demod(y, None)
        fm_check(got, a_ref, np.abs(y[1:] * np.conj(y[:-1])))
    else:
        assert rel_err(got, y) < FIR_TOL


@pytest.mark.parametrize("kern", ["auto", "ab"])
@pytest.mark.parametrize("fm", [True, False])
def test_u8_ingest_mfma_interior_tiles(dd, fm, kern, select_kernel):
    """raw u8 input through the M = 1 kernels (255 taps: the overlap-save FFT kernel by default, FM or complex64 output; forced
    "ab": the MFMA path's persistent kernel, whose interior tiles read and widen the bytes themselves, and the tile-per-
    workgroup kernel for a chunk on an odd sample), two chunks, against the oracle"""
    import ctypes as C
    select_kernel(None if kern == "auto" else kern)
    hip = dd.hip
    lib = hip.lib()
    fs = 2400000
    L1, L2 = 70001, 50000
    raw = O.synth_iq_fm(L1 + L2, fs, 43)
    d = hip.DevArray.from_host(raw.reshape(-1))
    taps = np.ascontiguousarray(O.win_hamming(255))
    h = C.c_void_p()
    hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), 255, hip.cycles_q64(25000.0, fs), 1,
                                  hip.DD_CHAIN_NCO | hip.DD_CHAIN_U8_INPUT | (hip.DD_CHAIN_FM if fm else 0)))
    outs, pos = [], 0
    for n in (L1, L2):
        no = lib.dd_chain_out_count(h, n)
        o = hip.DevArray(no, np.float32 if fm else np.complex64)
        hip.check(lib.dd_chain_process(h, d.ptr + 2 * pos, o.ptr, n, None, None))
        # first chunk (4-byte aligned bytes): the two-matrix-set kernel's u8 flavour (FM or complex output); the second
        # chunk starts on an odd sample (2-byte alignment): tile-per-workgroup kernel
        # (FM output with 255 taps: the overlap-save FFT kernel takes the aligned chunk, whole)
        # (round 4: the FFT kernel lays its block grid by the OUTPUT's alignment and takes any input alignment)
        # (round 5: Hamming 255 is the running-sum kernel's, FM or complex64 output)
        want = hip.DD_KERNEL_COS_RS if kern == "auto" else (hip.DD_KERNEL_MFMA_AB if (2 * pos) % 4 == 0 else hip.DD_KERNEL_MFMA_TILES)
        assert lib.dd_chain_last_kernel(h) == want
        outs.append(o.to_host())
        pos += n
    assert lib.dd_chain_path(h) == 1
    lib.dd_chain_destroy(h)
    got = np.concatenate(outs)
    y = O.FilterState(taps).applyOn(O.nco(O.grid_c64(raw), 25000.0, fs, 0))
    if fm:
        a_ref, _ = O.fm_demod(y, None)
        fm_check(got, a_ref, np.abs(y[1:] * np.conj(y[:-1])))
    else:
        assert rel_err(got, y) < FIR_TOL


def test_resident_raw_recording_through_commsignal(dd):
    """source.read_device_raw: views of the recording kept in HBM as raw uint8 pairs.  A commSignal built on one
    runs the fused chain with the u8 ingest flavour (equal to the chain on read()'s complex64 samples, chunked,
    state carried); anything that is not fused sees widened complex samples; odd (2-byte aligned) view starts work."""
    from directdemod_amd import source
    raw = O.synth_iq_fm(700001, 2048000, 31)
    src = source.IQarray(raw, 2048000)
    v = src.read_device_raw(0, 1000)
    assert v is not None and v.dtype == dd.hip.IQ8 and v.n == 1000
    # .signal of an untouched raw view == read()
    assert np.array_equal(dd.comm.commSignal(2048000, src.read_device_raw(12345, 20001)).signal, src.read(12345, 20001))
    res = []
    for reader in (src.read_device_raw, src.read):
        ck = dd.chunker.chunker(src, 200001)                    # odd chunk starts: views 2-byte aligned only
        flt, fm = dd.filters.blackmanHarris(151), dd.demod_fm.demod_fm()
        out = dd.comm.commSignal(2048000)
        for a, b in ck.getChunks:
            out.extend(dd.comm.commSignal(2048000, reader(a, b), ck).offsetFreq(30000.0).filter(flt)
                       .bwLim(60240, uniq="First").funcApply(fm.demod))
        res.append(np.asarray(out.signal))
    assert res[0].shape == res[1].shape and np.array_equal(res[0], res[1])
    # a zero-phase (not fused) filter and a lone offsetFreq on a raw view
    a = dd.comm.commSignal(2048000, src.read_device_raw(1001, 9001)).offsetFreq(25000.0).filter(dd.filters.hamming(31, zeroPhase=True)).signal
    b = dd.comm.commSignal(2048000, src.read(1001, 9001)).offsetFreq(25000.0).filter(dd.filters.hamming(31, zeroPhase=True)).signal
    assert np.array_equal(a, b)
    # limitData shifts the window the views come from
    src.limitData(5000, 300000)
    assert np.array_equal(dd.comm.commSignal(2048000, src.read_device_raw(0, 777)).signal, src.read(0, 777))
