"""
GPU parity of the audio-rate rows (R2 FFT resample, A1 Hilbert envelope, X1
normalised correlation, X2 peak pick, F2 filtfilt) and of the end-to-end configs 3
and 4, against the reference's golden vectors and the oracle.

Tolerances: float64 stages 1e-9 relative (FFT/summation order differs from
NumPy's); sync index picks and all lengths/rates are exact.
"""
import os

import numpy as np
import pytest

from oracle import dd_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dd():
    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()
    from directdemod_amd import _hip
    _hip.require_gpu()
    from directdemod_amd import comm, filters, demod_fm, demod_am, chunker, _ops, noaa_sync, source

    class NS:
        pass
    ns = NS()
    ns.hip, ns.comm, ns.filters, ns.demod_fm, ns.demod_am, ns.chunker, ns.ops, ns.noaa, ns.source = \
        _hip, comm, filters, demod_fm, demod_am, chunker, _ops, noaa_sync, source
    from directdemod_amd import constants
    ns.constants = constants
    return ns


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def rel_err(got, ref):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))


@pytest.fixture(params=[0, 1, 2])
def ops(request, golden_dir):
    return _load(golden_dir, "ops_seed%d.npz" % request.param)


def test_resample_strict_golden(dd, ops):
    ang = ops["fm_nostate"]
    s = dd.comm.commSignal(60235, ang).bwLim(40960, True)
    assert s.sampRate == 40960 and s.length == len(ops["resample_60235_40960"])
    assert rel_err(s.signal, ops["resample_60235_40960"]) < 1e-9
    s = dd.comm.commSignal(200000, ang).bwLim(11025, True)
    assert s.sampRate == 11025
    assert rel_err(s.signal, ops["resample_200000_11025"]) < 1e-9


@pytest.mark.parametrize("n,num", [(1000, 333), (1001, 500), (1000, 1000), (999, 1500), (1024, 2048), (7, 3)])
def test_resample_vs_oracle_shapes(dd, n, num):
    x = np.random.default_rng(n).standard_normal(n)
    got = dd.ops.resample_fft(dd.hip.DevArray.from_host(x), num).to_host()
    assert rel_err(got, O.resample_fft(x, num)) < 1e-9


def test_am_envelope_golden(dd, ops):
    ang = ops["fm_nostate"]
    L = len(ang) + 1
    am = dd.demod_am.demod_am()
    a = am.demod(ang[:L - 1 if (L - 1) % 2 == 0 else L - 2])
    assert rel_err(a, ops["am_env_full"]) < 1e-9
    b = am.demod(np.resize(ang, 3000)) if L >= 3000 else am.demod(ang[:750])
    assert rel_err(b, ops["am_env_3000"]) < 1e-9
    with pytest.raises(TypeError):
        am.demod(np.ones(8, dtype=np.complex64))


def test_am_envelope_blocks_vs_oracle(dd):
    x = np.random.default_rng(3).standard_normal(10007)
    got = dd.demod_am.demod_am().demod_blocks(x, 3000)
    assert rel_err(got, O.am_demod_blocks(x, 3000)) < 1e-9


def test_filtfilt_complex128_vs_oracle(dd):
    x = np.random.default_rng(5).standard_normal(2000) + 1j * np.random.default_rng(6).standard_normal(2000)
    d = dd.hip.DevArray.from_host(x.astype(np.complex128))
    got = dd.ops.filtfilt(O.win_blackmanharris(151), d).to_host()
    assert rel_err(got, O.filtfilt(O.win_blackmanharris(151), [1.0], x)) < 1e-10


@pytest.fixture(scope="module")
def noaa_inputs(golden_dir):
    g = _load(golden_dir, "noaa_c4.npz")
    raw = O.synth_apt_iq(float(g["dur"]), 2048000, seed=1)
    return g, raw


def test_xcorr_and_peaks_3s_golden(dd, noaa_inputs):
    g, raw = noaa_inputs
    # oracle audio/envelope (pinned to the reference's in the CPU suite) as the stage input
    audio, rate = O.audio_chain(lambda a, b: O.read_iq_u8(raw, a, b), len(raw), 2048000, 30000.0,
                                O.win_blackmanharris(151), 60000, audio_rate=40960, strict=False)
    am = O.am_demod_blocks(audio[:3 * rate])
    d = dd.hip.DevArray.from_host(am)
    needle = O.sync_needle(O.NOAA_SYNCA, rate)
    xc = dd.ops.xcorr_norm(d, needle).to_host()
    assert np.max(np.abs(xc[:20000] - g["xcorr_3s_syncA_head"])) < 1e-9
    assert np.max(np.abs(xc[-2000:] - g["xcorr_3s_syncA_tail"])) < 1e-9
    assert abs(np.sum(xc) - float(g["xcorr_3s_syncA_sum"])) < 1e-6
    pk = dd.ops.find_peaks(dd.hip.DevArray.from_host(xc), rate, len(needle))
    assert np.array_equal(pk, g["peaks_3s_syncA"])
    sig = dd.comm.commSignal(rate, am)
    ns = dd.noaa.noaa_sync(None, 0.0)
    assert np.array_equal(ns.correlate_and_find_peaks(sig, O.NOAA_SYNCA), g["peaks_3s_syncA"])
    assert np.array_equal(ns.correlate_and_find_peaks(sig, O.NOAA_SYNCB), g["peaks_3s_syncB"])


def test_crude_tail_in_one_call_equals_the_staged_route(dd, noaa_inputs):
    """dd_noaa_crude_tail (envelope by a real transform pair, prefix sums once, both needles, selection + threshold +
    candidates in one persistent launch with grid-wide barriers) against the stage-by-stage entry points on the same audio:
    identical index lists for both sync words (and equal to the reference's golden lists), envelope to 1e-12; odd lengths and
    a length below one block included (the last block of the chunker rule is ragged)."""
    g, raw = noaa_inputs
    audio, rate = O.audio_chain(lambda a, b: O.read_iq_u8(raw, a, b), len(raw), 2048000, 30000.0,
                                O.win_blackmanharris(151), 60000, audio_rate=40960, strict=False)
    needles = [O.sync_needle(O.NOAA_SYNCA, rate), O.sync_needle(O.NOAA_SYNCB, rate)]
    ns = dd.noaa.noaa_sync(None, 0.0)
    for n, dt in ((len(audio), np.float32), (len(audio) - 1, np.float64), (3 * rate, np.float32), (200001, np.float64)):
        a = np.ascontiguousarray(audio[:n].astype(dt))
        d = dd.hip.DevArray.from_host(a)
        res = dd.ops.crude_tail(d, rate, needles, want_env=True)
        assert res is not None
        (pa, pb), env = res
        sig = ns.envelope(dd.comm.commSignal(rate, a))
        env_ref = np.asarray(sig.signal)
        assert np.max(np.abs(env.to_host() - env_ref)) <= 1e-12 * np.max(env_ref), n
        assert np.array_equal(pa, ns.correlate_and_find_peaks(sig, O.NOAA_SYNCA)), n
        assert np.array_equal(pb, ns.correlate_and_find_peaks(sig, O.NOAA_SYNCB)), n
    # the reference's own lists for the first three seconds (float64 envelope of the oracle's audio)
    am = O.am_demod_blocks(audio[:3 * rate])
    (pa, pb), _ = dd.ops.crude_tail(dd.hip.DevArray.from_host(np.ascontiguousarray(audio[:3 * rate])), rate, needles)
    assert np.array_equal(pa, g["peaks_3s_syncA"]) and np.array_equal(pb, g["peaks_3s_syncB"])
    assert len(am) == 3 * rate


def test_noaa_prepare_builds_on_the_host_what_the_first_calls_take(dd):
    """dd_noaa_prepare (round 6: host work only -- closed forms + a radix-2 transform per Hilbert-kernel spectrum, kept on the host; the call
    that needs one uploads it): lengths no other test uses, prepared from a thread as noaa_sync does, then the block envelope
    (dd_am_envelope_f64: what dd_noaa_crude_tail runs first) against the oracle's block rule (decode_noaa.py:631-657) and one accurate window against the oracle's chain."""
    import threading
    rng = np.random.default_rng(77)
    block = 100002                                     # even: the split (even / odd) kernel of 2^17 points; ragged last block 23457: the plain one
    n = 3 * block + 23457
    width = 40001                                      # windows of 80 002 IQ samples -> 80 001 angles, cyclic length 2^18
    th = [threading.Thread(target=dd.hip.on_callers_device(dd.ops.noaa_prepare), args=(n, block, 0)),
          threading.Thread(target=dd.hip.on_callers_device(dd.ops.noaa_prepare), args=(0, block, 2 * width))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert dd.hip.lib().dd_noaa_prepare(n, block, 2 * width, None) == 0          # again: nothing left to build
    a = rng.standard_normal(n)
    env = dd.demod_am.demod_am().demod_blocks(dd.hip.DevArray.from_host(a), block)      # (the block envelope dd_noaa_crude_tail runs: same spectra)
    ref = O.am_demod_blocks(a, block)
    assert np.max(np.abs(env.to_host() - ref)) <= 1e-12 * np.max(ref)
    raw = O.synth_apt_iq(0.6, 2048000, seed=9)
    src = dd.source.IQarray(raw, 2048000)
    ns = dd.noaa.noaa_sync(src, 30000.0)
    st = 614400 - width
    idx, pks, tms = ns.accurate_windows([st], 2 * width, dd.constants.NOAA_SYNCA)
    i, h, t = O.accurate_sync_window(O.read_iq_u8(raw, st, st + 2 * width), 2048000, 30000.0, dd.constants.NOAA_SYNCA)
    assert idx[0] == i + st and abs(pks[0] - h) < 1e-4


def test_crude_tail_many_candidates(dd):
    """Audio without sync words (noise): the peak threshold (decode_noaa.py:723-726) lets a good part of the correlation values
    through -- more than the 24 576 candidates per needle that come back with the counters in the entry's one copy, so the rest is
    fetched in a second one; with still more than 65 536 the entry declines and the caller goes stage by stage.  (The candidates
    come back in index order: every wave counts those of its stretch, then writes them at the offset the counts before it give.)  Index lists
    against the staged route."""
    rate = 40960
    rng = np.random.default_rng(99)
    needles = [O.sync_needle(O.NOAA_SYNCA, rate), O.sync_needle(O.NOAA_SYNCB, rate)]
    ns = dd.noaa.noaa_sync(None, 0.0)
    seen = set()
    for n in (20 * rate + 17, 8 * rate + 3, 6 * rate, 5 * rate + 1, 4 * rate, 3 * rate + 5, 2 * rate, rate + 999):
        a = np.ascontiguousarray(rng.normal(0.0, 0.3, n))
        sig = ns.envelope(dd.comm.commSignal(rate, a))
        counts = []
        for nd in needles:                      # values above the threshold, from the staged correlation
            cor = np.sort(dd.ops.xcorr_norm(sig.device_signal, nd).to_host())
            K = int(2 * (n / rate)) + 2
            avgpk = np.mean(cor[-K:])
            counts.append(int(np.count_nonzero(cor > avgpk - 0.25 * (avgpk - np.mean(cor[:K])))))
        res = dd.ops.crude_tail(dd.hip.DevArray.from_host(a), rate, needles)
        if max(counts) > 65536:
            assert res is None, (n, counts)
            seen.add("declined")
            continue
        assert res is not None, (n, counts)
        pa, pb = res[0]
        assert np.array_equal(pa, ns.correlate_and_find_peaks(sig, O.NOAA_SYNCA)), (n, counts)
        assert np.array_equal(pb, ns.correlate_and_find_peaks(sig, O.NOAA_SYNCB)), (n, counts)
        seen.add("second copy" if max(counts) > 24576 else "one copy")
    assert "second copy" in seen, seen


def test_c4_crude_and_accurate_sync_indices_golden(dd, noaa_inputs):
    """config 4 end to end on the device: index lists identical to the reference's"""
    g, raw = noaa_inputs
    src = dd.source.IQarray(raw, 2048000)
    ns = dd.noaa.noaa_sync(src, 30000.0)
    sa, sb = ns.getCrudeSync()
    assert ns.crudeRate == int(g["audio_rate"]) == 60235
    assert np.array_equal(sa, g["crude_syncA"])
    assert np.array_equal(sb, g["crude_syncB"])
    assert ns.useful == int(g["useful"]) == 1
    (ia, pa, ta), (ib, pb, tb) = ns.getAccurateSync()
    assert np.array_equal(ia, g["acc_syncA"])
    assert np.array_equal(ib, g["acc_syncB"])
    assert np.max(np.abs(np.array(pa) - g["acc_syncA_pk"])) < 1e-4
    assert np.max(np.abs(np.array(pb) - g["acc_syncB_pk"])) < 1e-4
    assert np.max(np.abs(np.array(ta) - g["acc_syncA_time"])) < 1e-4


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_resample_chunk_list_equals_per_chunk_resample(dd, dtype):
    """dd_resample_fft_chunks (batched plans per (length, target) group) against the per-chunk entry and the oracle's
    scipy.signal.resample restatement: the C3 shape (chunks of 83886 / 83887 samples -> 4624) and mixed lengths"""
    rng = np.random.default_rng(8)
    lengths = [83887, 83886, 83886, 83887, 83886, 5000, 83887, 5001, 5000]
    nums = [int(11025 * n / 200000) for n in lengths]
    x = rng.standard_normal(sum(lengths)).astype(dtype)
    offs = np.concatenate([[0], np.cumsum(lengths)[:-1]])
    d = dd.hip.DevArray.from_host(x)
    out, out_off = dd.ops.resample_fft_chunks(d, offs, lengths, nums)
    got = out.to_host()
    assert got.shape == (sum(nums),)
    for i, (o, n, m) in enumerate(zip(offs, lengths, nums)):
        piece = x[o:o + n].astype(np.float64)
        one = dd.ops.resample_fft(dd.hip.DevArray.from_host(piece), m).to_host()
        assert rel_err(got[out_off[i]:out_off[i] + m], one) < 1e-12, i
        assert rel_err(got[out_off[i]:out_off[i] + m], O.resample_fft(piece, m)) < 1e-9, i


def test_resample_chirp_convolution_own_transform_equals_the_library(dd, monkeypatch):
    """The chirp-z resampler's cyclic convolution (length 2^17 for config 3's chunks, 2^18 for longer ones) as three launches of
    the float64 transform of csrc/dd_hconv_kernels.h (pre-multiply, spectrum product and post-multiply inside them; the default)
    against the same convolution through the FFT library (DD_CZT_OWN=0), and both against the oracle's scipy.signal.resample."""
    rng = np.random.default_rng(21)
    lengths = [83887, 83886, 150001, 83886, 150001, 70001]
    nums = [int(11025 * n / 200000) for n in lengths]
    x = rng.standard_normal(sum(lengths))
    offs = np.concatenate([[0], np.cumsum(lengths)[:-1]])
    d = dd.hip.DevArray.from_host(x)
    res = {}
    for own in ("0", None):
        monkeypatch.delenv("DD_CZT_OWN", raising=False)
        if own:
            monkeypatch.setenv("DD_CZT_OWN", own)
        out, out_off = dd.ops.resample_fft_chunks(d, offs, lengths, nums)
        res[own] = out.to_host()
    assert rel_err(res[None], res["0"]) < 1e-12
    for i, (o, n, m) in enumerate(zip(offs, lengths, nums)):
        assert rel_err(res[None][out_off[i]:out_off[i] + m], O.resample_fft(x[o:o + n], m)) < 1e-9, i


@pytest.mark.timeout(900)
def test_c4_at_bench_duration_index_lists_golden(dd, golden_dir):
    """config 4 at BENCH duration (60 s, "Pass = identical index lists"): crude and accurate sync over a 60 s
    synthetic APT recording resident in HBM, every index equal to the reference's own getCrudeSync / getAccurateSync run
    (tests/golden/noaa_c4_60s.npz, tools/gen_golden.py --c4-60s; decode_noaa.py:769-880)."""
    g = _load(golden_dir, "noaa_c4_60s.npz")
    raw = O.synth_apt_iq(float(g["dur"]), 2048000, seed=int(g["seed"]))
    src = dd.source.IQarray(raw, 2048000)
    ns = dd.noaa.noaa_sync(src, 30000.0)
    sa, sb = ns.getCrudeSync()
    assert ns.useful == int(g["useful"]) == 1
    assert np.array_equal(sa, g["crude_syncA"]) and np.array_equal(sb, g["crude_syncB"])
    assert len(sa) == 120 and len(sb) == 120
    (ia, pa, ta), (ib, pb, tb) = ns.getAccurateSync()              # batched windows, recording resident (uploaded by the crude pass)
    assert np.array_equal(ia, g["acc_syncA"]) and np.array_equal(ib, g["acc_syncB"])
    # and once more through a fresh object over the now-resident recording (what bench.py's side line times)
    ns2 = dd.noaa.noaa_sync(src, 30000.0)
    sa2, sb2 = ns2.getCrudeSync()
    (ia2, _, _), (ib2, _, _) = ns2.getAccurateSync()
    assert np.array_equal(sa2, sa) and np.array_equal(sb2, sb) and np.array_equal(ia2, ia) and np.array_equal(ib2, ib)
    # the stage-by-stage crude route (envelope(), correlate_and_find_peaks() as the reference's code calls them)
    sa3, sb3 = dd.noaa.noaa_sync(src, 30000.0).getCrudeSync(fused=False)
    assert np.array_equal(sa3, sa) and np.array_equal(sb3, sb)


def test_c4_audio_stage_vs_golden(dd, noaa_inputs):
    g, raw = noaa_inputs
    src = dd.source.IQarray(raw, 2048000)
    ns = dd.noaa.noaa_sync(src, 30000.0)
    aud = ns.audio(40960, False, chunkSize=5000000)      # several chunks: state carried on the device
    assert aud.sampRate == 60235
    a = aud.signal
    d = np.abs(np.angle(np.exp(1j * (a[:20000] - g["audio_3s_head"]))))
    assert np.max(d) < 2e-5 and np.median(d) < 2e-6


def test_c3_chain_golden(dd, golden_dir):
    """config 3: chunked remez-127 / M=50 / FM / FFT-resample to 11 025 S/s"""
    g = _load(golden_dir, "chain_c3.npz")
    L = int(g["L"])
    x = O.grid_c64(O.synth_iq_fm(L, 1e7, int(g["seed"]), f_carrier=250e3, f_mod=1e3, dev=5.0))

    class _Src:
        length = L
    ck = dd.chunker.chunker(_Src(), 8192)
    out = dd.comm.commSignal(11025)
    rz = dd.filters.remez(10000000, [[0, 100e3], [150e3, 4999999]], [1, 0], ntaps=127)
    assert np.max(np.abs(np.asarray(rz.getB) - g["taps_remez127"])) < 1e-12
    fm = dd.demod_fm.demod_fm()
    for a, b in ck.getChunks:
        s = dd.comm.commSignal(10000000, x[a:b], ck).offsetFreq(250000.0).filter(rz) \
            .bwLim(200000, uniq="First").funcApply(fm.demod).bwLim(11025, True)
        out.extend(s)
    assert out.sampRate == int(g["chain_c3_rate"]) == 11025
    assert out.length == len(g["chain_c3"])
    # the FFT resample spreads the float32 discriminator error over the chunk
    assert np.max(np.abs(out.signal - g["chain_c3"])) < 2e-5


def _class_chunk_loop(dd, rate, L, chunk, get, taps, M, f_off, fm_on, strict_rate, out_rate, one_by_one=False):
    """the reference's chunk loop (decode scripts: chunker -> commSignal(...).offsetFreq.filter.bwLim.funcApply.bwLim -> extend)"""
    class _Src:
        length = L
    ck = dd.chunker.chunker(_Src(), chunk)
    out = dd.comm.commSignal(out_rate)
    filt = dd.filters.filter(taps, 1, storeState=True)
    fm = dd.demod_fm.demod_fm()
    for a, b in ck.getChunks:
        s = dd.comm.commSignal(rate, get(a, b), ck)
        if f_off is not None:
            s.offsetFreq(f_off)
        s.filter(filt).bwLim(rate // M, uniq="First")
        if fm_on:
            s.funcApply(fm.demod)
        if strict_rate:
            s.bwLim(strict_rate, True)
        out.extend(s)
        if one_by_one:
            dd.comm.flush_all()                            # every chunk executes on its own: the chunk-by-chunk route on the same slices
    return out, filt


@pytest.mark.parametrize("case", ["c3_c64", "c3_u8", "c4_u8_no_resample", "iq_out", "ragged_last"])
def test_class_chunk_loop_over_a_resident_recording_is_one_launch(dd, case):
    """The drop-in classes on a device-resident recording: the chunk loop's per-chunk operations are recorded, and run as
    ONE chunk-list launch (+ one batched resample) when the output is first needed -- bit for bit what the same loop
    gives chunk by chunk."""
    rate, L, chunk, M, f_off, fm_on, strict, out_rate, K = {
        "c3_c64": (10000000, 1 << 20, 83887, 50, 250000.0, True, 11025, 11025, 127),
        "c3_u8": (10000000, 1 << 20, 83887, 50, 250000.0, True, 11025, 11025, 127),
        "c4_u8_no_resample": (2048000, 1 << 21, 250000, 34, None, True, None, 2048000 // 34, 151),
        "iq_out": (2048000, 1 << 20, 131072, 8, 30000.0, False, None, 256000, 65),
        "ragged_last": (2048000, (1 << 20) + 12345, 100003, 34, 30000.0, True, 40960, 40960, 151),
    }[case]
    raw = O.synth_iq_fm(L, rate, 5, f_carrier=f_off or 0.0, f_mod=1e3, dev=5.0)
    taps = O.win_blackmanharris(K)
    u8 = "u8" in case or case == "ragged_last"
    if u8:
        res = dd.hip.DevArray.from_host(np.ascontiguousarray(raw).reshape(-1), dtype=np.uint8)
        res = dd.hip.DevArray(L, dd.hip.IQ8, ptr=res.ptr, base=res)
        one = lambda a, b: _own(dd, res.view(a, b - a))
    else:
        x = O.grid_c64(raw)
        res = dd.hip.DevArray.from_host(x, dtype=np.complex64)
        one = lambda a, b: x[a:b]
    got, filt = _class_chunk_loop(dd, rate, L, chunk, lambda a, b: res.view(a, b - a), taps, M, f_off, fm_on, strict, out_rate)
    assert len(dd.comm._pending) > 1                       # nothing has run yet
    g = got.signal
    nchunks = len(O.chunk_list(L, chunk))
    # ONE launch for the whole list: k_chain_decim_w over the list as one chunk (even M in 8..64, round 5)
    assert filt._last_kernel() == dd.hip.decim_wave_kernel(K, M) and filt._launch_count() == 1
    ref, f2 = _class_chunk_loop(dd, rate, L, chunk, one, taps, M, f_off, fm_on, strict, out_rate)
    r = ref.signal
    assert f2._launch_count() == nchunks                   # (private copies: a launch per chunk)
    assert got.length == ref.length == len(g) == len(r) and got.sampRate == ref.sampRate
    assert g.dtype == r.dtype and np.array_equal(g, r)


def _own(dd, view):
    """a private device copy of a view (not a slice of a larger buffer: the classes then run it at once)"""
    d = dd.hip.DevArray(view.n, view.dtype)
    dd.hip.check(dd.hip.lib().dd_memcpy_d2d(d.ptr, view.ptr, view.n * view.dtype.itemsize, None), "d2d")
    return d


def test_class_chunk_loop_deferred_extend_keeps_the_reference_order_of_events(dd):
    """what the lazily extended container must not get wrong: a chunk signal changed after extend() contributes the samples
    it had AT extend(); reading the container in the middle of the loop, mixing in a chunk that cannot join the list, and
    extending with host signals in between all give the sequential result"""
    rate, L, chunk, M = 2048000, 600000, 100000, 34
    raw = O.synth_iq_fm(L, rate, 9, f_carrier=30000.0, f_mod=1e3, dev=5.0)
    x = O.grid_c64(raw)
    res = dd.hip.DevArray.from_host(x, dtype=np.complex64)
    taps = O.win_blackmanharris(151)

    def loop(get, meddle):
        class _Src:
            length = L
        ck = dd.chunker.chunker(_Src(), chunk)
        out = dd.comm.commSignal(rate // M)
        filt = dd.filters.filter(taps, 1, storeState=True)
        fm = dd.demod_fm.demod_fm()
        lens = []
        for i, (a, b) in enumerate(ck.getChunks):
            s = dd.comm.commSignal(rate, get(a, b), ck).offsetFreq(30000.0).filter(filt).bwLim(rate // M, uniq="First")
            s.funcApply(fm.demod)
            out.extend(s)
            if meddle:
                if i == 1:
                    s.updateSignal(np.zeros(5))                      # the container already has this chunk's samples
                if i == 2:
                    lens.append(float(np.sum(out.signal)))           # read in the middle of the loop
                if i == 3:
                    out.extend(dd.comm.commSignal(rate // M, np.arange(7.0)))      # a host signal in between
            else:
                if i == 2:
                    lens.append(float(np.sum(out.signal)))
                if i == 3:
                    out.extend(dd.comm.commSignal(rate // M, np.arange(7.0)))
        return out, lens

    got, gl = loop(lambda a, b: res.view(a, b - a), True)
    ref, rl = loop(lambda a, b: x[a:b], False)
    assert got.length == ref.length and gl == rl
    assert np.array_equal(got.signal, ref.signal)


def test_extend_adopts_the_contiguous_outputs_of_a_chunk_list(dd):
    """A chunk loop over a resident recording runs as ONE chunk-list call whose outputs lie back to back in one buffer; the empty
    container the chunks were extend()ed into then takes a VIEW of that stretch (no copy per chunk).  The shared samples must
    behave like the reference's copies (comm.py:163): growing the container afterwards, changing a chunk signal afterwards and
    dropping the chunk signals leave both sides intact."""
    rate, L, chunk, M = 2048000, 500000, 100000, 34
    x = O.grid_c64(O.synth_iq_fm(L, rate, 19, f_carrier=30000.0, f_mod=1e3, dev=5.0))
    res = dd.hip.DevArray.from_host(x, dtype=np.complex64)
    taps = O.win_blackmanharris(151)

    def loop(get):
        class _Src:
            length = L
        ck = dd.chunker.chunker(_Src(), chunk)
        out = dd.comm.commSignal(rate // M)
        filt = dd.filters.filter(taps, 1, storeState=True)
        fm = dd.demod_fm.demod_fm()
        sigs = []
        for a, b in ck.getChunks:
            s = dd.comm.commSignal(rate, get(a, b), ck).offsetFreq(30000.0).filter(filt).bwLim(rate // M, uniq="First")
            s.funcApply(fm.demod)
            out.extend(s)
            sigs.append(s)
        return out, sigs

    got, sigs = loop(lambda a, b: res.view(a, b - a))
    ref, _ = loop(lambda a, b: x[a:b])
    d = got.device_signal
    assert d._base is not None and d._base is sigs[0]._dev._base and d.ptr == sigs[0]._dev.ptr        # a view, nothing copied
    first = np.array(got.signal)
    assert np.array_equal(first, ref.signal)
    pieces = [np.array(s.signal) for s in sigs]
    got.extend(dd.comm.commSignal(rate // M, np.arange(5.0)))                  # growing copies into a buffer of the container's own
    sigs[1].updateSignal(np.zeros(3))                                           # a chunk signal replaced afterwards
    assert np.array_equal(np.asarray(got.signal)[:len(first)], first) and got.length == len(first) + 5
    assert np.array_equal(sigs[0].signal, pieces[0]) and np.array_equal(sigs[2].signal, pieces[2])
    del sigs
    import gc
    gc.collect()
    assert np.array_equal(np.asarray(got.signal)[:len(first)], first)


@pytest.mark.parametrize("which", ["strict", "freqs"])
def test_chunk_changed_after_extend_by_an_unrecorded_operation(dd, which):
    """extend() copies at call time in the reference (comm.py:163).  Here a chunk with pending operations is only NOTED by the
    container, so every operation that changes the chunk afterwards must first let the container take the samples the chunk had:
    the recorded operations do (through _record), and so must the two that run at once -- the strict bwLim behind anything but an
    FM demodulation (comm.py:110-116) and offsetFreq with a per-sample frequency array (decode_funcube.py:228)  (ADVICE r3)."""
    rate, L, chunk, M = 2048000, 300000, 100000, 34
    x = O.grid_c64(O.synth_iq_fm(L, rate, 19, f_carrier=30000.0, f_mod=1e3, dev=5.0))
    res = dd.hip.DevArray.from_host(x, dtype=np.complex64)
    taps = O.win_blackmanharris(151)

    def loop(meddle):
        class _Src:
            length = L
        ck = dd.chunker.chunker(_Src(), chunk)
        out = dd.comm.commSignal((rate // M) // 2 if which == "strict" else rate // M)
        filt = dd.filters.filter(taps, 1, storeState=True)
        fm = dd.demod_fm.demod_fm()
        for a, b in ck.getChunks:
            s = dd.comm.commSignal(rate, res.view(a, b - a), ck).offsetFreq(30000.0).filter(filt).bwLim(rate // M, uniq="First")
            if which == "strict":
                # real data whose last pending operation is NOT the demodulation: the strict bwLim then runs at once
                # (after an FM demodulation it is recorded with the chain)
                s.funcApply(fm.demod).bwLim(s.sampRate // 2, uniq="Second")
            out.extend(s)                                   # operations pending: noted, not copied
            if meddle and which == "strict":
                s.bwLim(s.sampRate // 2, True)
            elif meddle:
                keep = ck.get(dd.constants.CHUNK_FREQOFFSET)
                s.offsetFreq(np.full(s.length, 123.0))
                ck.set(dd.constants.CHUNK_FREQOFFSET, keep)          # (the meddling must not move the NEXT chunk's NCO index)
        return out

    got, ref = loop(True), loop(False)
    assert got.length == ref.length
    assert np.array_equal(got.signal, ref.signal)


@pytest.mark.parametrize("seed", range(12))
def test_class_chunk_loop_random_shapes_three_ways(dd, seed):
    """seeded random chunk loops (length, chunk size, taps, decimation, NCO on/off, FM on/off, strict resample on/off, u8 or
    complex64): the loop over slices of the resident recording run as one chunk-list launch == the same slices chunk by chunk,
    bit for bit; private copies of the chunks agree to rounding (a copy is aligned where a slice at an odd sample offset is not,
    which selects between the persistent and the plain decimating kernel: their NCO phasors round differently, found by seed 2);
    all agree with the oracle's chunk loop (float64) to the float32 chain's tolerance"""
    rng = np.random.default_rng(1000 + seed)
    rate = int(rng.choice([2048000, 2400000, 10000000]))
    M = int(rng.choice([2, 3, 8, 34, 50, 100]))
    K = int(rng.choice([2, 15, 64, 127, 151, 255, 300]))
    L = int(rng.integers(40000, 900000))
    chunk = int(rng.integers(max(3 * K, 5000), max(3 * K, 5000) + L // 2))
    use_nco = bool(rng.integers(0, 2))
    fm_on = bool(rng.integers(0, 4) > 0)
    strict = int(rng.choice([0, 11025, 40960])) if fm_on and rate // M > 41000 else 0
    u8 = bool(rng.integers(0, 2))
    f_off = float(rng.choice([25000.0, -30000.0, 250000.0])) if use_nco else None
    raw = O.synth_iq_fm(L, rate, 50 + seed, f_carrier=f_off or 10000.0, f_mod=1e3, dev=5.0)
    taps = O.firwin_lowpass(K, 0.4 / M) if K > 2 else np.array([0.5, 0.5])
    x = O.grid_c64(raw)
    if u8:
        base = dd.hip.DevArray.from_host(np.ascontiguousarray(raw).reshape(-1), dtype=np.uint8)
        res = dd.hip.DevArray(L, dd.hip.IQ8, ptr=base.ptr, base=base)
        private = lambda a, b: _own(dd, res.view(a, b - a))
    else:
        res = dd.hip.DevArray.from_host(x, dtype=np.complex64)
        private = lambda a, b: x[a:b]
    out_rate = strict if strict else rate // M
    got, f1 = _class_chunk_loop(dd, rate, L, chunk, lambda a, b: res.view(a, b - a), taps, M, f_off, fm_on, strict, out_rate)
    ref, f2 = _class_chunk_loop(dd, rate, L, chunk, lambda a, b: res.view(a, b - a), taps, M, f_off, fm_on, strict, out_rate, one_by_one=True)
    prv, f3 = _class_chunk_loop(dd, rate, L, chunk, private, taps, M, f_off, fm_on, strict, out_rate)
    g, r, pv = got.signal, ref.signal, prv.signal
    nchunks = len(O.chunk_list(L, chunk))
    case = dict(seed=seed, rate=rate, M=M, K=K, L=L, chunk=chunk, nco=f_off, fm=fm_on, strict=strict, u8=u8, nchunks=nchunks)
    if nchunks >= 2:
        wave = M % 2 == 0 and 8 <= M <= 64 and 2 <= K <= 256           # k_chain_decim_w's decimations; the others: k_chain_decim_multi
        assert f1._last_kernel() == (dd.hip.decim_wave_kernel(K, M) if wave else dd.hip.DD_KERNEL_DECIM_MULTI), case
        assert f1._launch_count() == 1 and f2._launch_count() == nchunks, case
    assert got.length == ref.length == len(g) and g.dtype == r.dtype and np.array_equal(g, r), case
    assert len(pv) == len(g) and np.max(np.abs(pv - g)) < (1e-5 if fm_on else 1e-4 * np.max(np.abs(g))), case
    if fm_on:
        want, want_rate = O.audio_chain(lambda a, b: x[a:b], L, rate, f_off or 0.0, taps, rate // M, strict or None, bool(strict),
                                        chunk_size=chunk, use_nco=use_nco)
        assert want_rate == got.sampRate and len(want) == len(g), case
        if strict:
            assert np.max(np.abs(g - want)) < 5e-5, case           # the resample spreads the discriminator's float32 error
        else:
            d = np.abs(np.angle(np.exp(1j * (g - want))))
            assert np.max(d) < 2e-4 and np.median(d) < 5e-6, (case, float(np.max(d)))


def test_source_readers(dd, tmp_path):
    raw = O.synth_iq_noise(5000, 8)
    f = tmp_path / "x.dat"
    raw.tofile(f)
    s = dd.source.IQdat(str(f), 2048000)
    assert s.length == 5000
    assert np.array_equal(s.read(10, 500), O.read_iq_u8(raw, 10, 500))
    assert np.array_equal(s.read_device(10, 500).to_host(), O.read_iq_u8(raw, 10, 500))
    with pytest.raises(ValueError):
        s.read(10, 6000)
    s.limitData(100, 600)
    assert s.length == 500 and np.array_equal(s.read(0, 5), O.read_iq_u8(raw, 100, 105))
    w = tmp_path / "x.wav"
    from _wav import write_iq_wav
    write_iq_wav(w, raw, 2400000)                                     # canonical 44-byte header
    sw = dd.source.IQwav(str(w))
    assert sw.sampFreq == 2400000 and sw.length == 5000
    assert np.array_equal(sw.read(0, 5000), O.grid_c64(raw))
    sa = dd.source.IQwavAlt(str(w))                                   # source.py:237-324: rate not read from the header
    assert sa.sampFreq == dd.constants.IQ_SDRSAMPRATE and sa.length == 5000
    assert np.array_equal(sa.read(7, 4000), O.read_iq_u8(raw, 7, 4000))
    assert dd.source.IQwavAlt(str(w), 1234567).sampFreq == 1234567


@pytest.mark.parametrize("how", ["host_read_default_chunk", "host_read_chunks_1e6", "device_raw_chunks_1e6"])
def test_config1_afsk_front_end_from_a_wav_golden(dd, golden_dir, tmp_path, how):
    """SURVEY 8d C1 in its stated shape (VERDICT r5 "what's missing" 3): a 2.4 MS/s 8-bit stereo IQ.wav whose NAME carries the centre
    frequency (main.py:167-173 takes the offset from it) -> source.IQwav -> chunker -> offsetFreq -> blackmanHarris(151) -> bwLim(22050)
    [M = 108: the tile kernels' path, above k_chain_decim_b's 64] -> extend; demod_fm over the whole (decode_afsk1200.py:67-94) -- against
    the FM output the REFERENCE produced from the same file (tests/golden/c1_afsk_front.npz, tools/gen_golden.py --c1).  The reference's
    one chunk (PROC_CHUNKSIZE = 2e7 > the file), chunks of 10^6 samples (its result does not depend on the chunking: carried state), and the
    same chunks as raw u8 pairs read in place on the device (widened inside the fused kernel)."""
    from _wav import write_iq_wav
    g = _load(golden_dir, "c1_afsk_front.npz")
    fs = int(g["fs"])
    name = "synth_20180101_120000Z_145825000Hz_IQ.wav"
    raw = O.synth_afsk_iq(int(g["n_bits"]), fs, int(g["seed"]), f_carrier=float(g["offset"]))
    path = tmp_path / name
    write_iq_wav(path, raw, fs)
    # what main.py does with the file name and -f 145835000 (main.py:167-173)
    centre = int([i for i in name.split("_") if i[-2:] == "Hz"][0][:-2])
    offset = 145835000 - centre
    assert offset == int(g["offset"])
    src = dd.source.IQwav(str(path))
    assert src.sampFreq == fs and src.length == raw.shape[0]
    ck = dd.chunker.chunker(src) if how == "host_read_default_chunk" else dd.chunker.chunker(src, 1000000)
    sig = dd.comm.commSignal(src.sampFreq)
    bh = dd.filters.blackmanHarris(151)
    fm = dd.demod_fm.demod_fm()
    for i in ck.getChunks:
        chunk = src.read_device_raw(*i) if how.startswith("device_raw") else src.read(*i)
        c = dd.comm.commSignal(src.sampFreq, chunk, ck)
        c.offsetFreq(offset)
        c.filter(bh)
        c.bwLim(int(g["bw"]))
        sig.extend(c)
    sig.funcApply(fm.demod)
    assert sig.sampRate == int(g["rate_out"]) and sig.length == int(g["n_out"])
    assert bh._last_kernel() in (dd.hip.DD_KERNEL_DECIM_PERSISTENT, dd.hip.DD_KERNEL_DECIM_TILES, dd.hip.DD_KERNEL_DECIM_MULTI)
    got = np.asarray(sig.signal, dtype=np.float64)
    # two-tier FM tolerance against the reference's float64 angles (every 4th, and the first and last 2048 in full)
    for a, b in ((got[:2048], g["head"]), (got[-2048:], g["tail"]), (got[::4], g["every4"])):
        d = np.abs(np.angle(np.exp(1j * (a - b))))
        assert np.max(d) < 2e-5 and np.median(d) < 2e-6, (how, float(np.max(d)), float(np.median(d)))


def test_streaming_ring_feeder_equals_one_shot(dd):
    """u8 source -> pinned ring -> side-stream H2D -> fused u8 chain, chunked == one shot"""
    from directdemod_amd import stream
    raw = O.synth_iq_fm(300000, 2048000, 12, f_carrier=30e3)
    src = dd.source.IQarray(raw, 2048000)
    taps = O.win_blackmanharris(151)
    for M in (34, 1):
        ref, r2 = O.audio_chain(lambda a, b: O.read_iq_u8(raw, a, b), len(raw), 2048000, 30000.0, taps,
                                2048000 // M if M > 1 else 2048000)
        # every staging route: pageable hipMemcpyAsync, the recording's pages pinned in place (hipHostRegister windows:
        # chunks of 70001 samples end in the middle of a page, so consecutive windows meet inside one), pinned slots
        for staging, chunk in (("direct", 70001), ("registered", 70001), ("registered", 1000), ("pinned", 70001)):
            out, rate = stream.stream_fm_chain(src, taps, 30000.0, M, chunk_size=chunk, depth=3, staging=staging)
            got = out.to_host().astype(np.float64)
            assert rate == r2 and got.shape == ref.shape, (M, staging)
            d = np.abs(np.angle(np.exp(1j * (got - ref))))
            assert np.max(d) < 2e-5 and np.median(d) < 2e-6, (M, staging, np.max(d))


def test_iir_butter_golden(dd, ops):
    """F4 on the device: state carried over chunks (real and complex), plain, zero-phase"""
    from directdemod_amd import constants
    L = int(ops["L"])
    x = O.grid_c64(O.synth_iq_noise(L, int(ops["seed"])))
    cuts = ops["fir_cuts"]
    xr = x.real.astype(np.float64)
    bt = dd.filters.butter(60235, 4160.0)
    assert np.allclose(bt.getB, ops["iir_b"]) and np.allclose(bt.getA, ops["iir_a"])
    y = np.concatenate([bt.applyOn(xr[cuts[i]:cuts[i + 1]]) for i in range(3)])
    assert rel_err(y, ops["iir_lp_real_chunks"]) < 1e-8
    btc = dd.filters.butter(2048000, 20000.0)
    yc = np.concatenate([btc.applyOn(x[cuts[i]:cuts[i + 1]].astype(np.complex128)) for i in range(3)])
    # 20 kHz @ 2.048 MHz, order 6: poles within 3% of the unit circle -- the recurrence amplifies the
    # rounding differences between this fma ordering and SciPy's loop
    assert rel_err(yc, ops["iir_lp_cplx_chunks"]) < 1e-7
    yp = dd.filters.butter(60235, 1000.0, 3000.0, n=4, typeFlt=constants.FLT_BP, storeState=False).applyOn(xr)
    assert rel_err(yp, ops["iir_bp_plain"]) < 1e-7      # 8th-order band-pass recurrence: same remark
    yz = dd.filters.butter(60235, 4160.0, zeroPhase=True).applyOn(xr)
    assert rel_err(yz, ops["iir_lp_filtfilt"]) < 1e-7


# ----------------------------------------------------------------------------- AFSK1200 correlators (8f-4)
def test_afsk_correlators_golden(dd, golden_dir):
    """sign(binary_filter) and the bit-edge signal of the reference's own getMsg run, bit-exact"""
    from directdemod_amd import afsk
    g = _load(golden_dir, "afsk.npz")
    tb, spb = afsk.correlator_tables(int(g["bw"]))
    tb_o, spb_o = O.afsk_tables(int(g["bw"]))
    assert spb == spb_o == 18 and np.array_equal(tb, tb_o)
    bf = afsk.binary_filter(g["audio"], int(g["bw"]))
    assert bf.dtype == np.float64 and len(bf) == len(g["audio"])
    assert np.array_equal(np.sign(bf).astype(np.int8), g["sign"])
    assert np.array_equal(bf, O.afsk_binary_filter(g["audio"], tb_o))           # same operation order: same bits
    ch = afsk.bit_edges(bf, spb)
    assert np.array_equal(ch, g["edge_sums"].astype(np.float64) / spb)


@pytest.mark.parametrize("n,bw", [(18, 22050), (19, 22050), (5000, 22050), (100000, 44100), (4097, 48000)])
def test_afsk_correlators_vs_oracle(dd, n, bw):
    from directdemod_amd import afsk
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) * 0.3 + np.sin(2 * np.pi * 1700 * np.arange(n) / bw)
    tb, spb = O.afsk_tables(bw)
    bf = afsk.binary_filter(x, bw)
    assert np.array_equal(bf, O.afsk_binary_filter(x, tb))
    if n >= spb:
        assert np.array_equal(afsk.bit_edges(bf, spb), O.afsk_edges(bf, spb))
    else:
        with pytest.raises(ValueError):
            afsk.bit_edges(bf, spb)


def test_afsk_front_end_on_device(dd):
    """decode_afsk1200.getMsg's stage order (:67-158) end to end on the device, device arrays
    handed from stage to stage: fused NCO/BH151/decimate chain, FM, butter band-pass, correlators."""
    from directdemod_amd import afsk, constants
    fs, bw = 22050 * 40, 22050
    raw = O.synth_afsk_iq(96, fs, 11)
    x = O.grid_c64(raw)
    sig = dd.comm.commSignal(fs, x).offsetFreq(0).filter(dd.filters.blackmanHarris(151)).bwLim(bw) \
        .funcApply(dd.demod_fm.demod_fm().demod)
    sig.filter(dd.filters.butter(sig.sampRate, 1200 - 500, 2200 + 500, typeFlt=constants.FLT_BP))
    audio = np.asarray(sig.signal, dtype=np.float64)
    # oracle: same stages in float64
    y = O.FilterState(O.win_blackmanharris(151)).applyOn(O.nco(x, 0.0, fs, 0))
    y, rate, _, _ = O.decimate_carry(y, fs, bw, 0)
    a, _ = O.fm_demod(y, None)
    import scipy.signal as ss
    b_, a_ = ss.butter(6, [700 / (0.5 * rate), 2700 / (0.5 * rate)], btype="bandpass")
    ref_audio = ss.lfilter(b_, a_, a, zi=ss.lfilter_zi(b_, a_))[0]
    assert sig.sampRate == rate and len(audio) == len(ref_audio)
    assert np.max(np.abs(audio - ref_audio)) < 5e-4 * np.max(np.abs(ref_audio))   # f32 chain in front of a resonant band-pass
    tb, spb = O.afsk_tables(bw)
    bf = afsk.binary_filter(audio, bw)
    bf_ref = O.afsk_binary_filter(ref_audio, tb)
    strong = np.abs(bf_ref) > 0.05 * np.max(np.abs(bf_ref))
    assert np.array_equal(np.sign(bf[strong]), np.sign(bf_ref[strong]))           # bit decisions agree wherever they are decisions


# ----------------------------------------------------------------------------- F4 at IQ rate: block-parallel IIR (8f-3)
@pytest.mark.parametrize("cplx", [False, True])
def test_iir_block_parallel_long_inputs(dd, cplx):
    """Inputs above 4096 samples take the block-parallel recurrence (block end states, two-level
    scan, re-run); the carried state must hand over between it and the one-lane kernel in
    either direction.  Checker: scipy.signal.lfilter, the routine the reference calls."""
    import scipy.signal as ss
    from directdemod_amd import constants
    rng = np.random.default_rng(5)
    n = 300000
    x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
    x = x.astype(np.complex128 if cplx else np.float64)
    for args, kw, tol in (((60235, 4160.0), {}, 1e-9),
                          ((2048000, 20000.0), {}, 1e-7),
                          ((22050, 700.0, 2700.0), {"typeFlt": constants.FLT_BP}, 1e-8)):
        f = dd.filters.butter(*args, **kw)
        b, a = np.asarray(f.getB), np.asarray(f.getA)
        zi = ss.lfilter_zi(b, a)
        cuts = [0, 70001, 70001 + 3000, 70001 + 3000 + 131072, n]      # parallel, one-lane, parallel (whole blocks), parallel (ragged)
        got = np.concatenate([f.applyOn(x[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)])
        ref = ss.lfilter(b, a, x, zi=zi.astype(x.dtype))[0]
        assert got.dtype == x.dtype
        assert rel_err(got, ref) < tol, (args, rel_err(got, ref))
        # stateless form
        f0 = dd.filters.butter(*args, storeState=False, **kw)
        assert rel_err(f0.applyOn(x[:100000]), ss.lfilter(b, a, x[:100000])) < tol


@pytest.mark.parametrize("tail_blocks", [1, 5, 14, 31])
def test_iir_wave_kernel_partial_last_workgroup(dd, tail_blocks):
    """complex128 input takes the one-wave LDS-DMA block passes (32 blocks per workgroup); its write pass counts on 16
    stores per step, which the last workgroup does not issue when it owns fewer than 32 blocks or a ragged last block:
    nb % 32 in {1, 5, 14, 31} with a partial last block, against scipy.signal.lfilter"""
    import scipy.signal as ss
    rng = np.random.default_rng(40 + tail_blocks)
    n = 256 * (32 * 40 + tail_blocks) - 100
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex128)
    f = dd.filters.butter(2400000, 100000.0, storeState=False)
    y = f.applyOn(dd.hip.DevArray.from_host(x))
    ref = ss.lfilter(np.asarray(f.getB), np.asarray(f.getA), x)
    assert rel_err(y.to_host(), ref) < 1e-9


def test_iir_block_parallel_complex64_iq(dd):
    """decode_funcube.py:160 shape: complex64 IQ through a butter low-pass, device array in and out"""
    import scipy.signal as ss
    x = O.grid_c64(O.synth_iq_fm(1 << 20, 2400000, 3))
    f = dd.filters.butter(2400000, 100000.0, storeState=False)
    d = dd.hip.DevArray.from_host(x)
    y = f.applyOn(d)
    assert isinstance(y, dd.hip.DevArray) and y.dtype == np.complex128 and y.n == len(x)
    ref = ss.lfilter(np.asarray(f.getB), np.asarray(f.getA), x.astype(np.complex128))
    assert rel_err(y.to_host(), ref) < 1e-9


@pytest.mark.parametrize("n", [3000, 4096, 65536 + 33, (1 << 20) + 1, (1 << 25) + 4097])
def test_iir_complex64_input_read_in_place_equals_the_widened_route(dd, n):
    """dd_iir_c64 (round 6): complex64 samples read by the block passes as they are (k_iir_blocks_w32) -- bit for bit what dd_iir_f64 gives
    on the widened copy (same recurrence, same block states), 1e-9 against lfilter; odd lengths (the last 16-byte pair holds one sample),
    a ragged last workgroup, both block lengths (256 below 2^25 samples, 1024 from there), the short in-place route below 4096, and the
    state carried over two chunks (filters.py:75 storeState)."""
    import scipy.signal as ss
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    f = dd.filters.butter(2048000, 20000.0, storeState=False)
    d = dd.hip.DevArray.from_host(x)
    y = f.applyOn(d)
    assert y.dtype == np.complex128 and y.n == n
    wide = dd.hip.DevArray.from_host(x.astype(np.complex128))
    yw = dd.filters.butter(2048000, 20000.0, storeState=False).applyOn(wide)
    got = y.to_host()
    assert np.array_equal(got, yw.to_host())
    if n <= (1 << 20) + 1:
        # (a sixth-order low-pass at 1 % of the sample rate, the reference's Funcube shape: poles within 0.03 of the unit circle -- the float64
        #  recurrence with fused multiply-adds and SciPy's without differ by 6e-9 of the peak here, 1e-9 holds for the wider filters above)
        ref = ss.lfilter(np.asarray(f.getB), np.asarray(f.getA), x.astype(np.complex128))
        assert rel_err(got, ref) < 1e-7
    # two chunks with the state carried: equal to one call over both
    cut = (n // 3) | 1
    g = dd.filters.butter(2048000, 20000.0)                 # storeState defaults to True (filters.py:239)
    a = g.applyOn(dd.hip.DevArray.from_host(x[:cut])).to_host()
    b = g.applyOn(dd.hip.DevArray.from_host(x[cut:])).to_host()
    g2 = dd.filters.butter(2048000, 20000.0)
    whole = g2.applyOn(wide).to_host()
    assert np.max(np.abs(np.concatenate([a, b]) - whole)) <= 1e-7 * np.max(np.abs(whole))      # (other block boundaries: the conditioning above)


# ----------------------------------------------------------------------------- X1: both forms of the correlation
@pytest.mark.parametrize("kind", ["runs_even", "runs_odd", "dense", "many_runs"])
def test_xcorr_norm_forms_vs_oracle(dd, kind):
    """piecewise-constant needles take the prefix-sum form, anything else the direct sum; both
    against the oracle's definition (np 'same' centring for even and odd needle lengths)"""
    rng = np.random.default_rng(17)
    n = 50000
    h = np.abs(rng.standard_normal(n)) + 0.2
    if kind == "runs_even":
        needle = O.sync_needle(O.NOAA_SYNCA, 60235) if hasattr(O, "NOAA_SYNCA") else ((np.repeat(rng.integers(0, 2, 40), 14) * 233) + 11) / 255
    elif kind == "runs_odd":
        needle = ((np.repeat(rng.integers(0, 2, 39), 17) * 233) + 11) / 255
    elif kind == "dense":
        needle = rng.standard_normal(301)
    else:
        needle = np.repeat(rng.standard_normal(200), 2)        # 200 runs of 2: more runs than the fast path takes
    d = dd.hip.DevArray.from_host(h)
    got = dd.ops.xcorr_norm(d, needle).to_host()
    ref = O.xcorr_norm(h, needle)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref))


def test_xcorr_norm_silent_stretch(dd):
    """windows of exact zeros: the reference's formula (decode_noaa.py:673) divides rounding noise of its
    FFT correlation by zero there (inf or NaN); the device forms give NaN.  Non-finite in the same places,
    equal everywhere else."""
    h = np.abs(np.random.default_rng(3).standard_normal(20000)) + 0.1
    h[5000:9000] = 0.0
    needle = ((np.repeat(np.array([0, 0, 1, 1, 0, 0, 1, 1, 0, 0]), 30) * 233) + 11) / 255
    got = dd.ops.xcorr_norm(dd.hip.DevArray.from_host(h), needle).to_host()
    ref = O.xcorr_norm(h, needle, exact_energy=True)
    bad_ref = ~np.isfinite(ref)
    assert bad_ref.sum() > 3000 and np.array_equal(~np.isfinite(got), bad_ref)
    assert np.max(np.abs(got[~bad_ref] - ref[~bad_ref])) < 1e-9


# ----------------------------------------------------------------------------- demod_amFLT (demod_am.py:35-62)
@pytest.mark.parametrize("cplx", [False, True])
def test_demod_amflt_vs_scipy(dd, cplx):
    """butter low-pass of |sig| with the state carried over chunks; short (one-lane) and long
    (block-parallel) chunks; real and complex input"""
    import scipy.signal as ss
    rng = np.random.default_rng(9)
    n = 60000
    car = np.exp(2j * np.pi * 0.11 * np.arange(n)) * (1 + 0.5 * np.sin(2 * np.pi * 0.001 * np.arange(n)))
    sig = car + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    sig = sig if cplx else sig.real
    dem = dd.demod_am.demod_amFLT(60235, 4160.0)
    cuts = [0, 3000, 50000, n]
    got = np.concatenate([dem.demod(sig[cuts[i]:cuts[i + 1]]) for i in range(3)])
    b, a = ss.butter(6, 4160.0 / (0.5 * 60235))
    ref = ss.lfilter(b, a, np.abs(sig), zi=ss.lfilter_zi(b, a))[0]
    assert got.dtype == np.float64 and rel_err(got, ref) < 1e-9
    d = dd.hip.DevArray.from_host(sig.astype(np.complex64) if cplx else sig)
    out = dd.demod_am.demod_amFLT(60235, 4160.0).demod(d)
    assert isinstance(out, dd.hip.DevArray)
    ref32 = ss.lfilter(b, a, np.abs(sig.astype(np.complex64)) if cplx else np.abs(sig), zi=ss.lfilter_zi(b, a))[0]
    assert rel_err(out.to_host(), ref32) < (1e-6 if cplx else 1e-9)


def test_iir_block_parallel_long_block_length(dd):
    """from 2^25 samples the block-parallel IIR switches to 1024-sample blocks (its own double-double block map);
    state carried into a following short chunk"""
    import scipy.signal as ss
    rng = np.random.default_rng(12)
    n = (1 << 25) + 12345
    x = rng.standard_normal(n + 50000)
    f = dd.filters.butter(2048000, 20000.0)
    b, a = np.asarray(f.getB), np.asarray(f.getA)
    got = np.concatenate([f.applyOn(x[:n]), f.applyOn(x[n:])])
    ref = ss.lfilter(b, a, x, zi=ss.lfilter_zi(b, a))[0]
    assert rel_err(got, ref) < 1e-7


@pytest.mark.parametrize("cplx", [False, True])
def test_iir_three_level_scan_short_blocks(dd, cplx):
    """3 000 001 samples: 256-sample blocks, 184 groups -> the scan goes through super-groups; odd length
    (the staged block kernels fetch 16-byte units: the last sample starts a unit of its own)"""
    import scipy.signal as ss
    rng = np.random.default_rng(21)
    n = 3000001
    x = rng.standard_normal(n + 7000) + (1j * rng.standard_normal(n + 7000) if cplx else 0)
    x = x.astype(np.complex128 if cplx else np.float64)
    f = dd.filters.butter(2048000, 20000.0)
    b, a = np.asarray(f.getB), np.asarray(f.getA)
    got = np.concatenate([f.applyOn(x[:n]), f.applyOn(x[n:])])
    ref = ss.lfilter(b, a, x, zi=ss.lfilter_zi(b, a).astype(x.dtype))[0]
    assert rel_err(got, ref) < 1e-7


def test_afsk_front_end_matches_the_reference_run(dd, golden_dir):
    """config 1's route (decode_afsk1200.py:67-98: offsetFreq -> blackmanHarris(151) -> bwLim(22050) -> demod_fm ->
    butter band-pass) on the device against the audio the reference itself produced for the same synthetic
    recording (tests/golden/afsk.npz, captured from its getMsg run), and the bit decisions behind it"""
    from directdemod_amd import afsk, constants
    g = _load(golden_dir, "afsk.npz")
    fs, bw = int(g["fs_iq"]), int(g["bw"])
    raw = O.synth_afsk_iq(int(g["n_bits"]), fs, int(g["seed"]))
    src = dd.source.IQarray(raw, fs)
    ck = dd.chunker.chunker(src)
    sig = dd.comm.commSignal(fs)
    bhf, fm = dd.filters.blackmanHarris(151), dd.demod_fm.demod_fm()
    for a, b in ck.getChunks:
        c = dd.comm.commSignal(fs, src.read(a, b), ck)
        c.offsetFreq(0)
        c.filter(bhf)
        c.bwLim(bw)
        sig.extend(c)
    sig.funcApply(fm.demod)
    sig.filter(dd.filters.butter(sig.sampRate, 1200 - 500, 2200 + 500, typeFlt=constants.FLT_BP))
    audio = np.asarray(sig.signal, dtype=np.float64)
    ref = g["audio"]
    assert sig.sampRate == bw and audio.shape == ref.shape
    assert np.max(np.abs(audio - ref)) < 5e-4 * np.max(np.abs(ref))          # float32 chain in front of a resonant band-pass
    bf = afsk.binary_filter(audio, bw)
    bf_ref = O.afsk_binary_filter(ref, O.afsk_tables(bw)[0])
    strong = np.abs(bf_ref) > 0.05 * np.max(np.abs(bf_ref))
    assert np.array_equal(np.sign(bf[strong]).astype(np.int8), g["sign"][strong])


# ---------------------------------------------------------------- zero-phase FIR, tiled kernels
@pytest.mark.parametrize("n", [1477, 2048, 2049, 4095, 6000, 118151])
def test_filtfilt_tiled_real_vs_oracle(dd, n):
    """hamming(492) zero-phase on float64 at lengths around the 2048-output tiles (and the shortest legal one)"""
    x = np.random.default_rng(n).standard_normal(n) + 0.5
    got = dd.filters.hamming(492, zeroPhase=True).applyOn(x)
    ref = O.filtfilt(O.win_hamming(492), [1.0], x)
    assert rel_err(got, ref) < 1e-12


@pytest.mark.parametrize("k", [1, 7, 8, 9, 151, 152])
def test_filtfilt_tiled_tap_counts(dd, k):
    """tap counts on both sides of the 8-tap chunks of the tiled kernel; complex64 (f32 arithmetic) and float64"""
    rng = np.random.default_rng(100 + k)
    taps = rng.standard_normal(k)
    x = rng.standard_normal(5000)
    assert rel_err(dd.ops.filtfilt(taps, dd.hip.DevArray.from_host(x)).to_host(), O.filtfilt(taps, [1.0], x)) < 1e-11
    xc = (rng.standard_normal(5000) + 1j * rng.standard_normal(5000)).astype(np.complex64)
    got = dd.ops.filtfilt(taps, dd.hip.DevArray.from_host(xc)).to_host()
    assert got.dtype == np.complex64
    assert rel_err(got, O.filtfilt(taps, [1.0], xc.astype(np.complex128))) < 2e-5


def test_filtfilt_long_filter_uses_plain_kernels(dd):
    """a filter too long for the LDS tile still runs (one lane per output)"""
    rng = np.random.default_rng(9)
    taps = rng.standard_normal(6000) / 6000
    x = rng.standard_normal(20000)
    assert rel_err(dd.ops.filtfilt(taps, dd.hip.DevArray.from_host(x)).to_host(), O.filtfilt(taps, [1.0], x)) < 1e-10


# ---------------------------------------------------------------- accurate-sync windows, batched
def test_accurate_sync_batched_equals_per_window(dd, noaa_inputs, monkeypatch):
    g, raw = noaa_inputs
    src = dd.source.IQarray(raw, 2048000)
    ns = dd.noaa.noaa_sync(src, 30000.0)
    one = ns.getAccurateSync(batched=False)
    gathered = ns.getAccurateSync(batched=True, resident=False)      # windows gathered on the host, uploaded per batch
    for (i1, p1, t1), (i2, p2, t2) in zip(one, gathered):
        assert np.array_equal(i1, i2) and np.max(np.abs(np.array(p1) - np.array(p2))) < 1e-9
    # default batch and batches smaller than the window count; the envelope as one real convolution (default)
    # and through the library's length-N transforms
    for batch, hilbert in ((None, None), ("3", None), (None, "fft"), (None, "lib"), ("3", "lib")):
        monkeypatch.delenv("DD_SYNC_BATCH", raising=False)
        monkeypatch.delenv("DD_SYNC_HILBERT", raising=False)
        if batch:
            monkeypatch.setenv("DD_SYNC_BATCH", batch)
        if hilbert:
            monkeypatch.setenv("DD_SYNC_HILBERT", hilbert)
        many = ns.getAccurateSync(batched=True)
        for (i1, p1, t1), (i2, p2, t2) in zip(one, many):
            assert len(i1) >= 2 and np.array_equal(i1, i2)
            assert np.max(np.abs(np.array(p1) - np.array(p2))) < 1e-9
            assert [v is None for v in t1] == [v is None for v in t2]
            a = np.array([v for v in t1 if v is not None])
            b = np.array([v for v in t2 if v is not None])
            assert np.max(np.abs(a - b)) < 1e-9 * np.max(np.abs(a))
    # golden index lists of the reference through both forms
    assert np.array_equal(one[0][0], g["acc_syncA"]) and np.array_equal(many[1][0], g["acc_syncB"])


@pytest.mark.parametrize("L,nwin", [(118152, 5), (65537, 2), (131072, 1), (100001, 4), (65536, 3), (32769, 2), (50001, 1)])
def test_sync_envelope_three_launch_transform(dd, L, nwin):
    """The envelope stage of the accurate-sync windows alone (decode_noaa.py:852 -> demod_am.py:29, abs(hilbert(x)) of each
    window's FM audio): the 512 x 512 float64 transform of csrc/dd_hconv_kernels.h (two windows per complex image, three
    launches) against the FFT library's padded real transforms on the same device-side audio (agreement at rounding level),
    and against scipy.signal.hilbert on the host (the float32 discriminator differs in the last bit there).  Odd window counts
    (a pair with an empty second half), the shortest and the longest window the 2^18 padding (rows of 512) and the 2^17 padding
    (rows of 256: four rows per wave) take."""
    import ctypes as C
    import scipy.signal
    rng = np.random.default_rng(L + nwin)
    # a wandering phase with a wandering amplitude: audio with structure at every scale, like the filtered IQ of a window
    ph = np.cumsum(rng.normal(0, 0.3, (nwin, L)) + 0.2 * np.sin(np.arange(L) * 2 * np.pi / 493.0), axis=1)
    x = ((1.0 + 0.5 * np.sin(np.arange(L) / 37.0)) * np.exp(1j * ph)).astype(np.complex64)
    d_x = dd.hip.DevArray.from_host(x.reshape(-1))
    envs = []
    for route in (0, 1):
        d_e = dd.hip.DevArray(nwin * (L - 1), np.float64)
        dd.hip.check(dd.hip.lib().dd_debug_sync_envelope(d_x.ptr, L, nwin, route, d_e.ptr, None), "dd_debug_sync_envelope")
        envs.append(d_e.to_host().reshape(nwin, L - 1))
    scale = np.max(envs[1])
    assert np.max(np.abs(envs[0] - envs[1])) <= 1e-12 * scale
    audio = np.angle(x[:, 1:].astype(np.complex128) * np.conj(x[:, :-1].astype(np.complex128)))
    ref = np.abs(scipy.signal.hilbert(audio, axis=1))
    assert np.max(np.abs(envs[0] - ref)) <= 2e-6 * scale


def test_accurate_sync_both_words_in_one_call(dd):
    """dd_noaa_sync_windows_multi (both window lists of getAccurateSync in one call, a needle index per window) against one
    call per sync word: identical picks; heights to rounding (two windows share one complex transform in the envelope stage, so a
    window's last bits depend on its partner).  Window counts that leave the last pair half empty."""
    raw = O.synth_apt_iq(1.4, 2048000, seed=11)
    src = dd.source.IQarray(raw, 2048000)
    ns = dd.noaa.noaa_sync(src, 30000.0)
    ns.getCrudeSync()                                                   # (puts the recording into HBM)
    width = int(3 * dd.constants.NOAA_T * len(dd.constants.NOAA_SYNCA) * 2048000)
    sa = [int(v) for v in np.linspace(0, src.length - 2 * width, 7)]
    sb = [int(v) for v in np.linspace(1234, src.length - 2 * width - 77, 4)]
    both = ns.accurate_windows([sa, sb], 2 * width, [dd.constants.NOAA_SYNCA, dd.constants.NOAA_SYNCB])
    each = [ns.accurate_windows(sa, 2 * width, dd.constants.NOAA_SYNCA), ns.accurate_windows(sb, 2 * width, dd.constants.NOAA_SYNCB)]
    for (i1, h1, t1), (i2, h2, t2) in zip(both, each):
        assert len(i1) == len(i2) and np.array_equal(i1, i2)
        assert np.max(np.abs(np.array(h1) - np.array(h2))) < 1e-12
        assert [v is None for v in t1] == [v is None for v in t2]
    # an empty list beside a full one
    only_b = ns.accurate_windows([[], sb], 2 * width, [dd.constants.NOAA_SYNCA, dd.constants.NOAA_SYNCB])
    assert len(only_b[0][0]) == 0 and np.array_equal(only_b[1][0], each[1][0])


def test_accurate_sync_windows_at_half_the_sample_rate(dd, monkeypatch):
    """A recording at 1.024 MS/s: the search windows are 59 076 samples, the envelope's cyclic convolution pads to 2^17 and runs
    through the rows-of-256 form of the own transform (four rows per wave).  Picks, heights and post-sync means against the
    library route (DD_SYNC_HILBERT=lib) and, for the picks, against the oracle's per-window chain."""
    fs = 1024000
    raw = O.synth_apt_iq(1.6, fs, seed=14)
    src = dd.source.IQarray(raw, fs)
    ns = dd.noaa.noaa_sync(src, 30000.0)
    width = int(3 * dd.constants.NOAA_T * len(dd.constants.NOAA_SYNCA) * fs)
    assert 32768 < 2 * width <= 65536
    starts = [int(v) for v in np.linspace(0, src.length - 2 * width, 5)]
    res = {}
    for hm in ("lib", None):
        monkeypatch.delenv("DD_SYNC_HILBERT", raising=False)
        if hm:
            monkeypatch.setenv("DD_SYNC_HILBERT", hm)
        res[hm] = ns.accurate_windows(starts, 2 * width, dd.constants.NOAA_SYNCA)
    assert np.array_equal(res["lib"][0], res[None][0])
    assert np.max(np.abs(np.array(res["lib"][1]) - np.array(res[None][1]))) < 1e-10
    ta, tb = res["lib"][2], res[None][2]
    assert [v is None for v in ta] == [v is None for v in tb]
    assert all(abs(a - b) <= 1e-10 * abs(a) for a, b in zip(ta, tb) if a is not None)
    for w in (0, 2, 4):
        a = starts[w]
        i, h, t = O.accurate_sync_window(O.read_iq_u8(raw, a, a + 2 * width), fs, 30000.0, dd.constants.NOAA_SYNCA)
        assert res[None][0][w] == i + a and abs(res[None][1][w] - h) < 1e-4


def test_accurate_sync_front_end_fused_into_the_first_filter_pass(dd, monkeypatch):
    """The windows' front end (uint8 pairs -> complex64, oscillator restarting at 0 per window) computed where the zero-phase
    filter's first pass stages its samples (k_filtfilt_tile mode 3, the default) against the front end as a kernel of its own
    writing the [windows][samples] array first: the same float32 operations on the same values, so every result is equal
    bit for bit.  Windows at both ends of the recording (the odd extension evaluates two samples per staged element there)."""
    raw = O.synth_apt_iq(1.3, 2048000, seed=12)
    src = dd.source.IQarray(raw, 2048000)
    ns = dd.noaa.noaa_sync(src, 30000.0)
    ns.getCrudeSync()
    width = int(3 * dd.constants.NOAA_T * len(dd.constants.NOAA_SYNCA) * 2048000)
    st = [0, 1, 54321, src.length - 2 * width - 1, src.length - 2 * width]
    res = {}
    for fr in ("kernel", None):
        monkeypatch.delenv("DD_SYNC_FRONT", raising=False)
        if fr:
            monkeypatch.setenv("DD_SYNC_FRONT", fr)
        res[fr] = ns.accurate_windows(st, 2 * width, dd.constants.NOAA_SYNCA)
    assert np.array_equal(res["kernel"][0], res[None][0]) and res["kernel"][1] == res[None][1] and res["kernel"][2] == res[None][2]


def test_accurate_sync_windows_oracle_chain(dd):
    """one window against the oracle's restatement of decode_noaa.py:852-853 (float64 SciPy chain)"""
    raw = O.synth_apt_iq(1.2, 2048000, seed=3)
    src = dd.source.IQarray(raw, 2048000)
    ns = dd.noaa.noaa_sync(src, 30000.0)
    width = int(3 * dd.constants.NOAA_T * len(dd.constants.NOAA_SYNCA) * 2048000)
    starts = [1024000 - width, 1024000 - width + 5000]
    idx, pks, tms = ns.accurate_windows(starts, 2 * width, dd.constants.NOAA_SYNCA)
    for w, a in enumerate(starts):
        i, h, t = O.accurate_sync_window(O.read_iq_u8(raw, a, a + 2 * width), 2048000, 30000.0, dd.constants.NOAA_SYNCA)
        assert idx[w] == i + a
        assert abs(pks[w] - h) < 1e-4
        assert (tms[w] is None) == (t is None)
        if t is not None:
            assert abs(tms[w] - t) < 1e-4


def test_accurate_sync_windows_rejects_long_windows(dd):
    raw = O.synth_apt_iq(0.6, 2048000, seed=4)
    src = dd.source.IQarray(raw, 2048000)
    ns = dd.noaa.noaa_sync(src, 30000.0)
    with pytest.raises(Exception, match="0.45 s"):
        ns.accurate_windows([0], 1000000, dd.constants.NOAA_SYNCA)


def test_accurate_sync_windows_c64_input_and_no_prefilter(dd):
    """dd_noaa_sync_windows called directly: complex64 windows give exactly what the raw uint8 pairs give, and
    without the envelope pre-filter (pre_ntaps = 0) the picks equal the per-stage route without `useFilter`"""
    import ctypes as C
    hip, lib = dd.hip, dd.hip.lib()
    raw = O.synth_apt_iq(1.2, 2048000, seed=5)
    src = dd.source.IQarray(raw, 2048000)
    width = int(3 * dd.constants.NOAA_T * len(dd.constants.NOAA_SYNCA) * 2048000)
    L = 2 * width
    starts = np.array([1024000 - width, 1024000 - width + 777, 400000], dtype=np.int64)
    bh = np.ascontiguousarray(dd.filters.blackmanHarris(151).getB, dtype=np.float64)
    pre = np.ascontiguousarray(dd.filters.hamming(492).getB, dtype=np.float64)
    needle = np.ascontiguousarray(dd.noaa.sync_needle(dd.constants.NOAA_SYNCA, 2048000), dtype=np.float64)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    d_u8 = hip.DevArray.from_host(raw.reshape(-1))
    d_c64 = hip.DevArray.from_host(src.read(0, src.length))

    def call(buf, kind, npre):
        pk, ht, ts = np.empty(3, np.int64), np.empty(3), np.empty(3)
        hip.check(lib.dd_noaa_sync_windows(buf.ptr, kind, starts.ctypes.data_as(ip), 3, L, hip.cycles_q64(30000.0, 2048000),
                                           bh.ctypes.data_as(dp), len(bh), pre.ctypes.data_as(dp), npre,
                                           needle.ctypes.data_as(dp), len(needle), 2048000.0,
                                           pk.ctypes.data_as(ip), ht.ctypes.data_as(dp), ts.ctypes.data_as(dp), None))
        return pk, ht, ts
    a, b = call(d_u8, 1, len(pre)), call(d_c64, 0, len(pre))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2], equal_nan=True)
    pk, ht, ts = call(d_u8, 1, 0)
    ns = dd.noaa.noaa_sync(src, 30000.0)
    for w, s0 in enumerate(starts):
        sig = dd.comm.commSignal(2048000, src.read_device(int(s0), int(s0) + L)).offsetFreq(30000.0) \
            .filter(dd.filters.blackmanHarris(151, zeroPhase=True)) \
            .funcApply(dd.demod_fm.demod_fm().demod).funcApply(dd.demod_am.demod_am().demod)
        p1, h1, t1 = ns.correlate_and_find_peaks(sig, dd.constants.NOAA_SYNCA, use_filter=False, extra=True)
        assert pk[w] == p1[0] and abs(ht[w] - h1[0]) < 1e-9
        assert (t1[0] is None) == bool(np.isnan(ts[w]))


@pytest.mark.parametrize("k", [1, 5, 37, 101, 492])
def test_real_fir_with_state_tiled(dd, k):
    """float64 FIR with the delay line carried from call to call (filters.py:64-70) through the LDS-tiled kernel:
    uneven chunks around the 2048-output tiles, one chunk shorter than the filter, against the oracle's lfilter"""
    rng = np.random.default_rng(300 + k)
    taps = rng.standard_normal(k)
    x = rng.standard_normal(12001)
    cuts = [0, 4097, 4097 + 20, 4097 + 20 + 2048, 12001]
    f = dd.filters.filter(taps, [1.0])
    got = np.concatenate([f.applyOn(x[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)])
    ref = O.FilterState(taps)
    want = np.concatenate([ref.applyOn(x[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)])
    assert got.dtype == np.float64
    assert rel_err(got, want) < 1e-12


def test_buffer_pool_fences_reuse_across_streams(dd):
    """a pooled buffer freed while another stream may still use it must not be handed to a new
    owner before that work is done.  While a side stream is registered, freed buffers are parked with one
    event per live stream; the next owner waits for them.  Without side streams no events are taken.
    The data check: a long asynchronous device-to-device copy on the null stream out of a buffer that is
    freed right away, then a new owner (same size class) filled over a side stream -- the copy's destination
    must still hold the original contents."""
    hip = dd.hip
    L = hip.lib()
    hip.pool_trim(0)
    n = 1 << 26                                            # 64 MiB
    a = hip.DevArray(n, np.uint8)
    ptr = a.ptr
    a.free()
    size = hip._pool_round(n)
    assert hip._pool[size][-1] == (ptr, [])                # only the null stream in use: nothing to fence
    b = hip.DevArray(n, np.uint8)
    assert b.ptr == ptr
    s = hip.stream_create()
    try:
        src = np.full(n, 7, dtype=np.uint8)
        hip.check(L.dd_memcpy_h2d(b.ptr, src.ctypes.data, n, None), "h2d")
        hip.sync()
        dst = hip.DevArray(n, np.uint8)
        for _ in range(4):                                 # asynchronous work on the null stream that reads b
            hip.check(L.dd_memcpy_d2d(dst.ptr, b.ptr, n, None), "d2d")
        b.free()
        ent = hip._pool[size][-1]
        assert ent[0] == ptr and len(ent[1]) == 2          # null stream + the side stream
        c = hip.DevArray(n, np.uint8)                      # waits for the fence before it returns
        assert c.ptr == ptr
        other = np.full(n, 9, dtype=np.uint8)
        hip.check(L.dd_memcpy_h2d(c.ptr, other.ctypes.data, n, s), "h2d on the side stream")
        hip.check(L.dd_stream_sync(s), "sync")
        hip.sync()
        assert np.all(dst.to_host() == 7)
        assert np.all(c.to_host() == 9)
    finally:
        hip.stream_destroy(s)
    assert not hip._streams
    d = hip.DevArray(16, np.uint8)
    d.free()
    assert hip._pool[hip._pool_round(16)][-1][1] == []


@pytest.mark.parametrize("n,fs,ft", [(83886, 200000, 11025), (5000, 48000, 44100), (4096, 1000, 3000), (300, 147, 160)])
def test_polyphase_resampler_stream_equals_scipy_resample_poly(dd, n, fs, ft):
    """EXTENSION row (north_star "polyphase resample", config 3): dd_rpoly_* against SciPy's resample_poly (the
    routine the stage is defined by) and the oracle's stream form: one shot + flush, and uneven chunks with the
    state carried on the device.  float64, 1e-12."""
    import scipy.signal as ss
    from directdemod_amd import resample
    x = np.random.default_rng(n).standard_normal(n)
    want = ss.resample_poly(x, ft, fs)
    assert np.max(np.abs(O.resample_poly(x, ft, fs) - want)) < 1e-12
    rs = resample.polyResampler(fs, ft)
    got = np.concatenate([rs.applyOn(x), rs.flush()])
    assert got.shape == want.shape and np.max(np.abs(got - want)) < 1e-12 * max(1.0, np.max(np.abs(want)))
    rs = resample.polyResampler(fs, ft)
    ors = O.PolyResampler(ft, fs) if n <= 5000 else None            # (the oracle's stream form is a Python loop per output)
    cuts = sorted({0, 1, n // 7, n // 7 + 3, n // 2, n - 2, n})
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        p = rs.applyOn(x[a:b])
        if ors is not None:
            assert p.shape == ors.applyOn(x[a:b]).shape             # the same outputs become available with every chunk
        parts.append(p)
    parts.append(rs.flush())
    got = np.concatenate(parts)
    assert got.shape == want.shape and np.max(np.abs(got - want)) < 1e-12 * max(1.0, np.max(np.abs(want)))
    rs.reset()                                                      # a new stream through the same object
    again = np.concatenate([rs.applyOn(x), rs.flush()])
    assert np.array_equal(again, np.concatenate([resample.polyResampler(fs, ft).applyOn(x), np.zeros(0)])) or again.shape == want.shape
    assert np.max(np.abs(again - want)) < 1e-12 * max(1.0, np.max(np.abs(want)))


def test_comm_resample_poly_in_a_chunk_loop(dd):
    """config 3's tail through the drop-in surface: FM audio at 200 kS/s -> 11 025 S/s, chunk by chunk, equals the
    one-shot polyphase resample of the whole audio (the reference's per-chunk FFT resample cannot: border effects)."""
    import scipy.signal as ss
    from directdemod_amd import resample
    fs, ft, n = 200000, 11025, 60000
    audio = np.sin(2 * np.pi * 1000 * np.arange(n) / fs) + 0.1 * np.random.default_rng(3).standard_normal(n)
    rs = resample.polyResampler(fs, ft)
    out = dd.comm.commSignal(ft)
    for a in range(0, n, 16384):
        out.extend(dd.comm.commSignal(fs, audio[a:a + 16384]).resamplePoly(rs))
    out.extend(dd.comm.commSignal(ft, rs.flush()))
    want = ss.resample_poly(audio, ft, fs)
    assert out.sampRate == ft and out.length == len(want) == -(-n * 441 // 8000)
    assert np.max(np.abs(np.asarray(out.signal) - want)) < 1e-12
    with pytest.raises(TypeError):
        dd.comm.commSignal(48000, audio[:100]).resamplePoly(rs)
