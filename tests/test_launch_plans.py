"""
Host arithmetic of two launch paths, checked without a GPU through the library's diagnostic entries (the library loads and these
two functions run on any machine: they never touch the device).

* dd_debug_fft1k_plan: where k_chain_fft1k lays its 768-output blocks over a chunk (base: the block grid follows the alignment of
  `out`) and which interior blocks a wave takes in which round (DDFft1kMap).  The kernel evaluates
      start(w, k) = 1 + wstart[k] + w r0[k] + c(w) (r1[k] - r0[k]),  len(w, k) = r1[k] if wave w owns b + 1 blocks else r0[k],
      c(w) = floor(NI w / NW) - b w
  and every interior block 1 .. nblk - 2 must come out exactly once, in contiguous runs, with every wave owning floor or ceil of
  NI / NW blocks (a wrong table here is silently wrong samples on the GPU, found only by the full-size parity tests).
* dd_debug_cos_fit: which tap sets the accurate-sync windows' zero-phase filter treats as a cosine series (prefix-sum form).
"""
import ctypes as C

import numpy as np
import pytest

from directdemod_amd import _hip


def _plan(L, s, align, ncu, rounds=0):
    out = (C.c_int * (7 + 3 * 32))()
    _hip.check(_hip.lib().dd_debug_fft1k_plan(L, s, align, ncu, rounds, out), "dd_debug_fft1k_plan")
    v = list(out)
    maxk = v[6]
    assert maxk == 32
    return dict(base=v[0], nblk=v[1], grid=v[2], nwaves=v[3], K=v[4], b=v[5], r0=v[7:7 + maxk], r1=v[7 + maxk:7 + 2 * maxk],
                wstart=v[7 + 2 * maxk:7 + 3 * maxk])


def _wave_runs(p, w):
    ni, nw, b = max(0, p["nblk"] - 2), p["nwaves"], p["b"]
    w0, w1 = (ni * w) // nw, (ni * (w + 1)) // nw
    big = (w1 - w0) > b
    c = w0 - b * w
    runs = []
    for k in range(p["K"]):
        r0, r1 = p["r0"][k], p["r1"][k]
        start = 1 + p["wstart"][k] + w * r0 + c * (r1 - r0)
        ln = r1 if big else r0
        if ln > 0:
            runs.append((start, ln))
    return w1 - w0, runs


@pytest.mark.parametrize("L,s,align,ncu,rounds", [
    (1 << 26, 1, 0, 256, 0), (1 << 26, 0, 0, 256, 0), (1 << 26, 1, 0, 256, 1), (1 << 26, 0, 5, 256, 0), ((1 << 26) + 12345, 1, 15, 256, 7),
    (1 << 22, 1, 0, 256, 0), (5000000, 0, 3, 256, 0), (3 * 768 + 5, 1, 0, 256, 0), (770, 0, 0, 256, 0), (1, 1, 0, 256, 0), (1, 0, 9, 256, 0),
    (2, 1, 0, 256, 0), (255, 0, 0, 256, 0), (100000, 1, 7, 4, 0), (100000, 1, 7, 4, 32), (100000, 0, 0, 1, 3), (40_000_000, 1, 0, 304, 0)])
def test_fft1k_plan_covers_every_block_once(L, s, align, ncu, rounds):
    p = _plan(L, s, align, ncu, rounds)
    base, nblk, nw = p["base"], p["nblk"], p["nwaves"]
    # the block grid: covers every stored output, starts a 64-byte line of `out`, the last block holds output L - 1
    assert base <= s and base > -768
    assert (base - s + align) % 16 == 0                       # out[base - s] sits `align + base - s` elements behind a line start
    assert base + 768 * nblk >= L and base + 768 * (nblk - 1) <= L - 1
    assert nblk >= 1 and nw >= 1 and nw == 4 * p["grid"] and p["grid"] <= 3 * ncu
    assert 1 <= p["K"] <= 32
    if rounds:
        assert p["K"] == min(rounds, p["b"] + 1, 32)
    ni = max(0, nblk - 2)
    seen = np.zeros(nblk, dtype=np.int32)
    for w in range(nw):
        own, runs = _wave_runs(p, w)
        assert own in (ni // nw, ni // nw + (1 if ni % nw else 0))
        assert sum(ln for _, ln in runs) == own
        for start, ln in runs:
            assert 1 <= start and start + ln <= nblk - 1, (w, start, ln)
            seen[start:start + ln] += 1
    assert np.all(seen[1:nblk - 1] == 1)
    assert seen[0] == 0 and (nblk == 1 or seen[nblk - 1] == 0)          # the edge blocks belong to wave 0 / the last wave


def test_fft1k_plan_rounds_tile_moving_windows():
    """the runs of one round, taken in wave order, tile one contiguous window of the stream; the windows follow each other"""
    p = _plan(1 << 26, 1, 0, 256)
    assert p["K"] == 4 and p["b"] == 28 and p["nwaves"] == 3072 and p["nblk"] == 87382
    pos = 1
    for k in range(p["K"]):
        for w in range(p["nwaves"]):
            own, _ = _wave_runs(p, w)
            ln = p["r1"][k] if own > p["b"] else p["r0"][k]
            if ln:
                big = own > p["b"]
                ni, nw = p["nblk"] - 2, p["nwaves"]
                c = (ni * w) // nw - p["b"] * w
                start = 1 + p["wstart"][k] + w * p["r0"][k] + c * (p["r1"][k] - p["r0"][k])
                assert start == pos, (k, w)
                pos += ln
                assert big in (True, False)
    assert pos == p["nblk"] - 1


def _win(kind, n):
    k = np.arange(n)
    if kind == "hamming":
        return 0.54 - 0.46 * np.cos(2 * np.pi * k / (n - 1))
    if kind == "bh":
        return 0.35875 - 0.48829 * np.cos(2 * np.pi * k / (n - 1)) + 0.14128 * np.cos(4 * np.pi * k / (n - 1)) - 0.01168 * np.cos(6 * np.pi * k / (n - 1))
    if kind == "hann":
        return 0.5 - 0.5 * np.cos(2 * np.pi * k / (n - 1))
    raise ValueError(kind)


@pytest.mark.parametrize("kind,n,Q,coef", [("hamming", 492, 1, [0.54, -0.46, 0, 0]), ("hamming", 255, 1, [0.54, -0.46, 0, 0]),
                                           ("bh", 151, 3, [0.35875, -0.48829, 0.14128, -0.01168]), ("hann", 101, 1, [0.5, -0.5, 0, 0])])
def test_cos_fit_recognises_the_reference_windows(kind, n, Q, coef):
    taps = np.ascontiguousarray(_win(kind, n), dtype=np.float64)
    a = (C.c_double * 4)()
    q = C.c_int(-1)
    assert _hip.lib().dd_debug_cos_fit(taps.ctypes.data_as(C.POINTER(C.c_double)), n, a, C.byref(q)) == 1
    assert q.value == Q
    assert np.allclose(list(a), coef, rtol=0, atol=1e-12)
    # the series reproduces the taps to rounding
    k = np.arange(n)
    fit = sum(a[i] * np.cos(2 * np.pi * i * k / (n - 1)) for i in range(4))
    assert np.max(np.abs(fit - taps)) <= 1e-13 * np.max(np.abs(taps))


def test_cos_fit_declines_everything_else():
    import scipy.signal as ss
    lib = _hip.lib()
    a = (C.c_double * 4)()
    q = C.c_int(-1)
    dp = C.POINTER(C.c_double)
    for taps in (ss.remez(127, [0, 100e3, 150e3, 4999999], [1, 0], fs=1e7),       # equiripple design: no cosine series
                 ss.windows.gaussian(151, 20.0),
                 np.ones(200),                                                    # a rolling average alone: not worth the kernel
                 _win("hamming", 32),                                             # too short
                 _win("hamming", 492) * (1 + 1e-9 * np.arange(492)),              # a window that is only nearly one
                 _win("hamming", 2000)):                                          # too long for the kernel's LDS window
        t = np.ascontiguousarray(taps, dtype=np.float64)
        assert lib.dd_debug_cos_fit(t.ctypes.data_as(dp), len(t), a, C.byref(q)) == 0


@pytest.mark.parametrize("L,s,align,ncu", [
    (1 << 26, 1, 0, 256), (1 << 26, 0, 0, 256), (1 << 26, 0, 5, 256), ((1 << 26) + 12345, 1, 15, 256), (1 << 22, 1, 0, 256), (5000000, 0, 3, 256),
    (3 * 1024 + 5, 1, 0, 256), (1025, 0, 0, 256), (1024, 1, 0, 256), (1, 1, 0, 256), (1, 0, 9, 256), (2, 1, 0, 256), (255, 0, 0, 256), (100000, 1, 7, 4),
    (40_000_000, 1, 0, 304)])
def test_cos1k_plan_lays_the_row_grid_by_the_outputs_alignment(L, s, align, ncu):
    """dd_debug_cos1k_plan (round 5): k_chain_cos1k's rows of 1024 samples.  out[base - s] must start a 64-byte line (`align` = how many
    elements `out` sits behind one), the rows must hold every sample from the first that has an angle (s) to L - 1 -- the one before it and
    the carried state belong to the row before the first run, which the kernel runs without stores -- and no wave may be left without a row."""
    out = (C.c_int * 4)()
    _hip.check(_hip.lib().dd_debug_cos1k_plan(L, s, align, ncu, out), "dd_debug_cos1k_plan")
    base, rows, grid, waves = list(out)
    assert (base - s + align) % 16 == 0 and base <= min(s, L - 1)
    assert base > min(s, L - 1) - 16                          # (the nearest such line start)
    assert rows >= 1 and base + 1024 * rows >= L and base + 1024 * (rows - 1) <= L - 1
    assert waves == 4 * grid and 1 <= grid <= 2 * ncu and waves - 4 < rows or rows < 4


@pytest.mark.parametrize("abs0,L,off,K,M,ncu", [
    (0, 1 << 26, 0, 151, 34, 256), (0, 1 << 26, 0, 127, 50, 256), (20000000, 20000000, 12, 151, 34, 256), (4194304 * 3, 1 << 22, 46, 127, 50, 256),
    (0, 5000, 0, 255, 8, 256), (123457, 70001, 7, 2, 8, 256), (999, 1, 0, 64, 64, 256), (0, 2047, 33, 256, 34, 304), (2047, 3, 1, 15, 10, 4)])
def test_decimw_plan_lays_rows_on_the_absolute_sample_grid(abs0, L, off, K, M, ncu):
    """dd_debug_decimw_plan (round 5): k_chain_decim_w's rows are the blocks of 2048 samples of the ABSOLUTE sample index; a chunk's rows run
    from the block of its first kept sample to the block of its last one; the window of every output starts on an even LDS sample (16-byte
    reads), one sample early behind a zero tap where the stream's phase makes it odd; decimations that are multiples of 8 get the padded image; and the chunks of a chunk loop (comm.py:123-125: the
    decimation phase follows on) see the same grid as the concatenation -- which is why a chunk list is one launch."""
    lib = _hip.lib()
    out = (C.c_int64 * 12)()
    Ld = len(range(off, L, M))
    _hip.check(lib.dd_debug_decimw_plan(abs0, Ld, K, M, off, ncu, out), "dd_debug_decimw_plan")
    R0, rows, phi, HP, e, K16, wpc, run_rows, bsum, NI, j1lo, img_got = list(out)
    first = abs0 + off
    assert phi == first % M
    if Ld == 0:
        assert rows == 0
        return
    last = first + (Ld - 1) * M
    assert 2048 * R0 <= first < 2048 * (R0 + 1) and 2048 * (R0 + rows - 1) <= last < 2048 * (R0 + rows)
    assert NI == -(-K // M) and bsum == (1 if NI <= 8 else 0)
    span = HP + 2048
    if bsum:
        # block sums (round 6, k_chain_decim_b): an output is the sum of NI <= 8 sums over blocks of M samples, a lane forms those of the
        # block that ends at its kept sample.  M samples in front of a row; a block starts on an even LDS sample (16-byte reads), one sample
        # early where the stream's phase makes it odd; M = 0 mod 4: two samples of gap between the blocks; partial sums 4 .. 7 have
        # non-zero taps from block sample j1lo on
        assert HP == M and e in (0, 1) and (HP + phi - M + 1 - e) % 2 == 0
        pad = M % 4 == 0
        img = ((span + 2 * (span // M + 4) + 32) & ~1) if pad else span + 32
        assert j1lo == max(0, 5 * M + e - K)
    else:
        assert HP % 2 == 0 and K - 1 <= HP <= K
        assert e in (0, 1) and (HP - K + 1 + phi - e) % 2 == 0           # (the offset of a kept sample in its block has phi's parity: 2048 and M are even)
        # M = 0 mod 8: the padded LDS image -- two samples of gap after every M of a window, zero taps over them
        pad = M % 8 == 0
        taps = K + e + (2 * ((K + e - 1) // M) if pad else 0)
        assert K16 % 16 == 0 and taps <= K16 < taps + 16
        img = ((span + 2 * (span // M + 4) + 2 * 16 + 40) & ~1) if pad else span + 16
    assert img == img_got
    assert 1 <= wpc <= 8 and wpc * 8 * (img + 32) <= 160 * 1024 and (wpc == 8 or (wpc + 1) * 8 * (img + 32) > 160 * 1024)
    assert 1 <= run_rows <= 8 or rows < run_rows * ncu * wpc
    # the next chunk of the loop: same phase, rows that follow on
    noff = (M - (L - off) % M) % M
    out2 = (C.c_int64 * 12)()
    _hip.check(lib.dd_debug_decimw_plan(abs0 + L, len(range(noff, 1 << 20, M)), K, M, noff, ncu, out2), "dd_debug_decimw_plan")
    assert out2[2] == phi and out2[4] == e and out2[0] in (R0 + rows - 1, R0 + rows) or out2[0] > R0 + rows


def test_decimw_plan_refuses_what_the_kernel_does_not_take():
    out = (C.c_int64 * 12)()
    for K, M in ((151, 33), (151, 6), (151, 66), (257, 34), (1, 34)):
        assert _hip.lib().dd_debug_decimw_plan(0, 1000, K, M, 0, 256, out) == _hip.DD_ERR_UNSUPPORTED


def test_a_variant_library_is_loaded_through_dd_lib_path_and_says_so(tmp_path):
    """measurement scripts no longer copy ablation builds over the product library; DD_LIB_PATH makes _hip load another build of
    the same C-ABI instead, with a line on stderr.  (Here: a copy of the product library under another name.)"""
    import os
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "directdemod_amd", "libdirectdemod_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    alt = str(tmp_path / "lib_variant.so")
    shutil.copy(lib, alt)
    env = dict(os.environ, DD_LIB_PATH=alt)
    r = subprocess.run([sys.executable, "-c", "from directdemod_amd import _hip; _hip.load(); print(_hip.LIB_PATH)"], capture_output=True, text=True,
                       env=env, cwd=root, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    assert r.stdout.strip() == alt
    assert "DD_LIB_PATH set, loading" in r.stderr and alt in r.stderr
