"""
Known-answer vectors transcribed from the reference's notebooks:
the only results the reference itself pins for this path.  They pin the oracle.
File:line citations are raw JSON line numbers of the .ipynb files under
/root/reference/experiments/.
"""
import numpy as np
import pytest

from oracle import dd_oracle as O


def test_rolling_average_stateless():
    # Experiment 3 - Filters - Initial tests.ipynb:131-132
    f = O.FilterState([0.5, 0.5], storeState=False)
    y = f.applyOn(np.arange(1, 20))
    assert np.allclose(y, np.arange(1, 20) - 0.5)
    assert y[0] == 0.5


def test_rolling_average_state_carry_and_unscaled_zi_quirk():
    # Experiment 3 ...ipynb:288, outputs :276-280 -- first sample is 1.0, not 0.5 (Q1)
    f = O.FilterState([0.5, 0.5], storeState=True)
    a = f.applyOn(np.arange(1, 10))
    b = f.applyOn(np.arange(10, 15))
    c = f.applyOn(np.arange(15, 20))
    assert np.allclose(a, [1.0, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5, 8.5])
    assert np.allclose(b, [9.5, 10.5, 11.5, 12.5, 13.5])
    assert np.allclose(c, [14.5, 15.5, 16.5, 17.5, 18.5])


def test_no_state_carry_border_values():
    # Experiment 3 ...ipynb:158,:164 -- what NOT carrying state looks like
    f = O.FilterState([0.5, 0.5], storeState=False)
    assert np.allclose(f.applyOn(np.arange(10, 15))[:2], [5.0, 10.5])
    assert np.allclose(f.applyOn(np.arange(15, 20))[:2], [7.5, 15.5])


def test_fm_discriminator_values():
    # Experiment 5 - FM demod.ipynb:72,77-79
    x = np.array([1 + 1j, 2 - 2j, 3 + 3j, 4 - 4j, 5 + 5j, 6 - 6j])
    y, _ = O.fm_demod(x, None, store_state=False)
    assert np.allclose(y, [-np.pi / 2, np.pi / 2, -np.pi / 2, np.pi / 2, -np.pi / 2])


def test_fm_one_sample_carry():
    # Experiment 5 ...ipynb:128-129,134-136: two 3-sample chunks -> 2 then 3 outputs
    x = np.array([1 + 1j, 2 - 2j, 3 + 3j, 4 - 4j, 5 + 5j, 6 - 6j])
    a, last = O.fm_demod(x[:3], None)
    b, last = O.fm_demod(x[3:], last)
    assert len(a) == 2 and len(b) == 3
    assert np.allclose(np.concatenate([a, b]), [-np.pi / 2, np.pi / 2, -np.pi / 2, np.pi / 2, -np.pi / 2])


def test_decimation_phase_carry():
    # Experiment 6 - Resampling and chunking.ipynb:99-103,114: range(100)@40 Hz,
    # chunks of 10, bwLim(10) with chunker == unchunked [0,4,...,96]
    x = np.arange(100)
    off = 0
    out = []
    for a, b in O.chunk_list(100, 10):
        y, rate, off, M = O.decimate_carry(x[a:b], 40, 10, off)
        out.append(y)
        assert rate == 10 and M == 4
    assert np.array_equal(np.concatenate(out), np.arange(0, 100, 4))
    # without the carry the result is wrong (:64)
    wrong = np.concatenate([x[a:b][0::4] for a, b in O.chunk_list(100, 10)])
    assert not np.array_equal(wrong, np.arange(0, 100, 4))


def _minus3db_hz(taps, fs):
    H = np.abs(np.fft.rfft(taps, 1 << 16))
    H = H / H[0]
    k = np.argmax(H < 10 ** (-3 / 20.0))
    return k * fs / (1 << 16)


def test_window_tap_design_minus_3db_points():
    # Experiment 4a:123 (BH 101 -> 20 kHz), 4b:116 (Hamming 101 -> 14 kHz),
    # 4c:116 (Gaussian 51, sigma 5 -> 56 kHz) at Fs = 2.048 MHz
    fs = 2.048e6
    assert abs(_minus3db_hz(O.win_blackmanharris(101), fs) - 20e3) < 2e3
    assert abs(_minus3db_hz(O.win_hamming(101), fs) - 14e3) < 2e3
    assert abs(_minus3db_hz(O.win_gaussian(51, 5), fs) - 56e3) < 4e3


def test_taps_are_raw_unnormalised_windows():
    # Hamming(255) sum 137.24; BH(151) sum 53.8126 (quirk Q2)
    assert abs(np.sum(O.win_hamming(255)) - 137.24) < 0.01
    assert abs(np.sum(O.win_blackmanharris(151)) - 53.8126) < 1e-3


def test_lfilter_zi_fir_is_history_of_ones():
    b = O.win_hamming(31)
    zi = O.lfilter_zi(b)
    assert np.allclose(zi, [np.sum(b[i + 1:]) for i in range(30)])
    x = np.random.default_rng(0).standard_normal(100) + 1j * np.random.default_rng(1).standard_normal(100)
    y1, _ = O.lfilter_fir(b, x, zi)
    y2, _ = O.fir_history_form(b, x, np.ones(30))
    assert np.allclose(y1, y2, atol=1e-12)


def test_chunker_rule():
    # chunker.py:36-45: exact multiple still ends with a full-size last chunk
    assert O.chunk_list(30, 10) == [[0, 10], [10, 20], [20, 30]]
    assert O.chunk_list(35, 10) == [[0, 10], [10, 20], [20, 30], [30, 35]]
    assert O.chunk_list(5, 10) == [[0, 5]]
    assert O.chunk_list(10, 10) == [[0, 10]]
    assert O.chunk_list(0, 10) == [[0, 0]]


def test_bwlim_rate_error():
    with pytest.raises(ValueError):
        O.decimate_carry(np.arange(10), 10, 40)


# ---------------------------------------------------------------------------------------------------------
# polyphase resampler (build-defined stage; the reference has none): the oracle's restatement of SciPy's
# published resample_poly is pinned against that routine itself, one shot and as a stream
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,up,down", [(1000, 3, 7), (83886, 11025, 200000), (2000, 2, 1), (777, 5, 5),
                                       (200, 441, 8000), (300, 160, 147), (1, 3, 2), (64, 1, 64)])
def test_resample_poly_restatement_equals_scipy(n, up, down):
    import scipy.signal as ss
    x = np.random.default_rng(n + up).standard_normal(n)
    want = ss.resample_poly(x, up, down)
    got = O.resample_poly(x, up, down)
    assert got.shape == want.shape and np.max(np.abs(got - want)) <= 1e-13 * max(1.0, np.max(np.abs(want)))
    if up != down and n * max(up, down) <= 2000000:      # (the stream form is a per-output Python loop: small cases only)
        rs = O.PolyResampler(up, down)
        cuts = sorted({0, n // 3, n // 3 + 1, (2 * n) // 3, n})
        parts = [rs.applyOn(x[a:b]) for a, b in zip(cuts[:-1], cuts[1:])] + [rs.flush()]
        st = np.concatenate(parts)
        assert st.shape == want.shape and np.max(np.abs(st - want)) <= 1e-13 * max(1.0, np.max(np.abs(want)))


def test_resample_poly_design_matches_scipy_filter():
    import scipy.signal as ss
    up, down, hp, npr = O.resample_poly_design(11025, 200000)
    assert (up, down) == (441, 8000)
    half_len = 10 * 8000
    h = ss.firwin(2 * half_len + 1, 1.0 / 8000, window=("kaiser", 5.0)) * up
    n_pre_pad = down - half_len % down
    assert npr == (half_len + n_pre_pad) // down and len(hp) == n_pre_pad + len(h)
    assert np.max(np.abs(hp[n_pre_pad:] - h)) < 1e-15
