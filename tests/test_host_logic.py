"""
Host-side logic of the drop-in classes (no GPU needed: the hot-path calls are only
recorded until a result is read): chunk lists, chunker variables, lengths, rates,
first-chunk quirks and the reference's error behaviour.
"""
import numpy as np
import pytest

from directdemod_amd import chunker, comm, constants, demod_fm, filters
from oracle import dd_oracle as O


class _Src:
    def __init__(self, n):
        self.length = n


@pytest.fixture(autouse=True)
def _clean_queue():
    comm._pending.clear()
    yield
    comm._pending.clear()


@pytest.mark.parametrize("n", [0, 1, 9, 10, 11, 20, 30, 35, 1000])
@pytest.mark.parametrize("cs", [1, 7, 10])
def test_chunk_list_matches_reference_rule(n, cs):
    assert chunker.chunker(_Src(n), cs).getChunks == O.chunk_list(n, cs)


def test_chunker_default_size_and_vars():
    ck = chunker.chunker(_Src(45000000))
    assert ck.getChunks == [[0, 20000000], [20000000, 40000000], [40000000, 45000000]]
    with pytest.raises(KeyError):
        ck.get("nope")
    assert ck.get("v", 5) == 5
    assert ck.get("v", 7) == 5
    ck.set("v", 9)
    assert ck.get("v") == 9


def test_constants_match_reference_values():
    assert constants.PROC_CHUNKSIZE == 20000000
    assert constants.CHUNK_FREQOFFSET == "freqoffset" and constants.CHUNK_BWLIM == "bwlim"
    assert constants.NOAA_SYNCA == O.NOAA_SYNCA and constants.NOAA_SYNCB == O.NOAA_SYNCB
    assert (constants.FLT_LP, constants.FLT_HP, constants.FLT_BP, constants.FLT_BS) == (0, 1, 2, 3)
    assert constants.NOAA_T == 1.0 / 4160 and constants.NOAA_MINPEAKDIST == 0.45


def test_ctor_errors():
    with pytest.raises(ValueError):
        comm.commSignal(0, np.zeros(4))
    with pytest.raises(TypeError):
        comm.commSignal(10, np.zeros((2, 2)))
    s = comm.commSignal(10.7, np.zeros(4))
    assert s.sampRate == 10 and s.length == 4


def test_ctor_copies_and_signal_is_internal():
    # Experiment 1: ctor copies; .signal returns the internal array (no copy)
    a = np.arange(5.0)
    s = comm.commSignal(10, a)
    a[0] = 99
    assert s.signal[0] == 0
    assert s.signal is s.signal


def test_bwlim_bookkeeping_with_chunker_matches_reference():
    # Experiment 6: range(100) @ 40 Hz, chunks of 10, bwLim(10): lengths, rate, phase var
    ck = chunker.chunker(_Src(100), 10)
    x = np.arange(100.0)
    off = 0
    for a, b in ck.getChunks:
        s = comm.commSignal(40, x[a:b], ck).bwLim(10)
        y, rate, off, M = O.decimate_carry(x[a:b], 40, 10, off)
        assert s.length == len(y) and s.sampRate == rate == 10
        assert ck.get(constants.CHUNK_BWLIM + "abcd") == off
    with pytest.raises(ValueError):
        comm.commSignal(10, x).bwLim(40)


def test_bwlim_rate_quirk_q7():
    s = comm.commSignal(2048000, np.zeros(3400, dtype=np.complex64)).bwLim(60000, uniq="First")
    assert s.sampRate == 60235 and s.length == 100


def test_offsetfreq_updates_chunker_index():
    ck = chunker.chunker(_Src(30), 10)
    for k, (a, b) in enumerate(ck.getChunks):
        comm.commSignal(100, np.zeros(b - a, dtype=np.complex64), ck).offsetFreq(5.0)
        assert ck.get(constants.CHUNK_FREQOFFSET) == b
    # a per-sample frequency array of the wrong length is the reference's broadcasting error (no GPU needed to say so)
    with pytest.raises(ValueError):
        comm.commSignal(100, np.zeros(4, dtype=np.complex64)).offsetFreq(np.zeros(3))


def test_fm_length_quirk_q3_at_call_time():
    fm = demod_fm.demod_fm()
    s1 = comm.commSignal(100, np.zeros(10, dtype=np.complex64)).funcApply(fm.demod)
    s2 = comm.commSignal(100, np.zeros(10, dtype=np.complex64)).funcApply(fm.demod)
    assert (s1.length, s2.length) == (9, 10)
    fm2 = demod_fm.demod_fm(storeState=False)
    s3 = comm.commSignal(100, np.zeros(10, dtype=np.complex64)).funcApply(fm2.demod)
    s4 = comm.commSignal(100, np.zeros(10, dtype=np.complex64)).funcApply(fm2.demod)
    assert (s3.length, s4.length) == (9, 9)


def test_chain_is_recorded_in_call_order():
    flt = filters.blackmanHarris(151)
    fm = demod_fm.demod_fm()
    ck = chunker.chunker(_Src(2000), 1000)
    sigs = []
    for a, b in ck.getChunks:
        sigs.append(comm.commSignal(2048000, np.zeros(b - a, dtype=np.complex64), ck)
                    .offsetFreq(30000.0).filter(flt).bwLim(60000, uniq="First").funcApply(fm.demod))
    assert [s.length for s in sigs] == [29, 29]           # 30 kept - 1 (first), then 29 kept
    assert comm._pending == sigs
    assert [op[0] for op in sigs[0]._ops] == ["nco", "fir", "decim", "fm"]
    assert sigs[1]._ops[0][2] == 1000 and sigs[1]._ops[2][2] == (34 - 1000 % 34) % 34


def test_extend_rate_rules():
    a = comm.commSignal(5)
    b = comm.commSignal(7, np.zeros(0))
    a.extend(b)                      # empty adopts the other's rate (comm.py:157-158)
    assert a.sampRate == 7
    c = comm.commSignal(9, np.zeros(0))
    a2 = comm.commSignal(5, np.zeros(0))
    a2.extend(c)
    assert a2.sampRate == 9


def test_filter_ctor_rules_and_errors():
    f = filters.hamming(255)
    assert np.allclose(f.getB, O.win_hamming(255)) and list(f.getA) == [1]
    assert np.allclose(filters.blackmanHarris(151).getB, O.win_blackmanharris(151))
    assert np.allclose(filters.gaussian(51, 5).getB, O.win_gaussian(51, 5))
    assert np.allclose(filters.rollingAverage(4).getB, [0.25] * 4)
    assert not filters.hamming(11, zeroPhase=True)._fusable()
    assert filters.hamming(11)._fusable()
    with pytest.raises(ValueError):
        filters.butter(1000, 100, typeFlt=constants.FLT_BP)
    with pytest.raises(ValueError):
        filters.butter(1000, 100, typeFlt=42)
    with pytest.raises(ValueError):
        filters.remez(1000, [], [])
    with pytest.raises(ValueError):
        filters.remez(1000, [[0, 100], [200, 500]], [1, 0])
    with pytest.raises(ValueError):
        filters.remez(1000, [[0, 100], [200, 400]], [1])
    r = filters.remez(10000000, [[0, 100e3], [150e3, 4999999]], [1, 0], ntaps=127)
    assert len(r.getB) == 127
    assert not filters.butter(2048000, 20000)._fusable()


def test_afsk_correlator_tables_match_the_oracle():
    from directdemod_amd import afsk
    from oracle import dd_oracle as O
    for bw in (22050, 44100, 48000):
        tb, spb = afsk.correlator_tables(bw)
        tb_o, spb_o = O.afsk_tables(bw)
        assert spb == spb_o and np.array_equal(tb, tb_o)


def test_iqwav_walks_riff_chunks_to_data(tmp_path):
    """source.py:66-68: the reference reads IQ.wav through scipy.io.wavfile.read, which finds the
    ``data`` chunk wherever it is; SDRSharp recordings (the file BASELINE config 1 names) carry an
    ``auxi`` chunk in front of it.  IQwav must return the same samples and rate as that call;
    IQwavAlt keeps the reference's fixed 44-byte offset (source.py:247)."""
    import scipy.io.wavfile
    from _wav import write_iq_wav
    from directdemod_amd import source
    raw = O.synth_iq_noise(3001, 5)
    plain, aux = tmp_path / "plain.wav", tmp_path / "sdrsharp.wav"
    write_iq_wav(plain, raw, 2400000)
    write_iq_wav(aux, raw, 2048000, before=[(b"auxi", bytes(range(164))), (b"LIST", b"INFOISFT\x05\0\0\0abcde")],
                 after=[(b"LIST", b"xyz")])
    for f, rate in ((plain, 2400000), (aux, 2048000)):
        ref_rate, ref_data = scipy.io.wavfile.read(str(f), True)                  # what the reference calls
        s = source.IQwav(str(f))
        assert s.sampFreq == ref_rate == rate and s.length == ref_data.shape[0] == 3001
        want = (ref_data[:, 0] + 1j * ref_data[:, 1]).astype("complex64") - (127.5 + 1j * 127.5)   # source.py:117-118
        assert np.array_equal(s.read(0, 3001), want)
        assert np.array_equal(s.read(17, 1200), O.read_iq_u8(raw, 17, 1200))
        assert np.array_equal(s.read_raw_u8(5, 9), raw[5:9].reshape(-1))
    assert source.IQwav(str(aux), 1234).sampFreq == 1234
    alt = source.IQwavAlt(str(plain))
    assert alt.length == 3001 and np.array_equal(alt.read(0, 3001), O.grid_c64(raw))
    with pytest.raises(ValueError):
        source.IQwav(str(aux)).read(0, 3002)
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"\0" * 100)
    with pytest.raises(ValueError):
        source.IQwav(str(bad))
    mono = tmp_path / "mono.wav"
    write_iq_wav(mono, raw, 8000, channels=1)
    with pytest.raises(TypeError):
        source.IQwav(str(mono))


def test_bench_apt_generator_equals_the_test_suites():
    """bench.py's C4 side line and tests/test_gpu_audio.py's 60 s golden test must run over the same recording"""
    import bench
    from oracle import dd_oracle as O
    assert np.array_equal(bench.synth_apt_iq(1.3, 2048000, seed=1), O.synth_apt_iq(1.3, 2048000, seed=1))


def test_star_import_brings_the_drop_in_modules():
    ns = {}
    exec("from directdemod_amd import *", ns)
    for m in ("comm", "filters", "demod_fm", "demod_am", "chunker", "constants", "source"):
        assert m in ns and hasattr(ns[m], "__name__"), m
    assert hasattr(ns["comm"], "commSignal") and hasattr(ns["filters"], "hamming") and hasattr(ns["chunker"], "chunker")
