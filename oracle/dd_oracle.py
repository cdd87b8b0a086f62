"""
CPU oracle for the DirectDemod per-sample hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``directdemod_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` use it, and there only as the checker / reported baseline, never
as the thing measured or shipped.

What it is: a NumPy (float64) restatement of the algorithm behind every row of
SURVEY.md §8(a).  The reference is pure Python and delegates its arithmetic to
SciPy/NumPy (un-vendored; pinned ``scipy==1.0.0``/``numpy==1.14.0`` in
``requirements.txt:3,11`` -- not installable here).  The SciPy routines the
reference calls (``lfilter``, ``lfilter_zi``, ``lfiltic``, ``filtfilt``,
``resample``, ``hilbert``, ``correlate``) are restated below from their published
definitions using NumPy only, so the oracle has no SciPy dependency.

How it is pinned (see tests/test_oracle_*.py, tools/gen_golden.py):
  * every known-answer vector printed in the reference's notebooks
    (experiments/Experiment 3/5/6) is reproduced;
  * golden vectors in tests/golden/*.npz were produced by importing the reference
    itself (``/root/reference``, with the 4-line SciPy/NumPy compat shim of
    SURVEY.md App. A) in the build container; the oracle matches them to 1e-9.

All citations ``file:line`` are relative to ``/root/reference/``.
"""
import math

import numpy as np

# ---------------------------------------------------------------------------
# constants the path reads (directdemod/constants.py:4-40)
# ---------------------------------------------------------------------------
PROC_CHUNKSIZE = 20000000               # constants.py:8
NOAA_T = 1.0 / 4160                     # constants.py:15
NOAA_SYNCA = [0, 0, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0,
              1, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]   # constants.py:16
NOAA_SYNCB = [0, 0, 0, 0, 1, 1, 1, 0, 0, 1, 1, 1, 0, 0, 1, 1, 1, 0, 0, 1,
              1, 1, 0, 0, 1, 1, 1, 0, 0, 1, 1, 1, 0, 0, 1, 1, 1, 0, 0, 0]   # constants.py:17
NOAA_PEAKHEIGHTWIGGLE = 0.25            # constants.py:18
NOAA_MINPEAKDIST = 0.45                 # constants.py:19
NOAA_DETECTMAXCHANGE = 5                # constants.py:21
NOAA_DETECTCONSSYNCSNUM = 10            # constants.py:22


# ---------------------------------------------------------------------------
# S1  source.read  (source.py:117-118, 209-210, 303-304)
# ---------------------------------------------------------------------------
def read_iq_u8(raw_iq_u8, a, b):
    """raw_iq_u8: uint8[N,2] (I,Q).  Returns complex64[b-a] = I + jQ - (127.5+127.5j)."""
    d = np.asarray(raw_iq_u8)
    samples = d[a:b, 0] + 1j * d[a:b, 1]                  # complex128 temp (source.py:117)
    return np.array(samples).astype("complex64") - (127.5 + 1j * 127.5)


# ---------------------------------------------------------------------------
# K1  chunker  (chunker.py:21-45)
# ---------------------------------------------------------------------------
def chunk_list(length, chunk_size=PROC_CHUNKSIZE):
    chunks = []
    i = 0
    while i + chunk_size < length:                         # chunker.py:36-38
        chunks.append([i, i + chunk_size])
        i += chunk_size
    if len(chunks) == 0:                                   # chunker.py:41-42
        chunks.append([0, length])
    elif chunks[-1][1] != length:                          # chunker.py:44-45
        chunks.append([chunks[-1][1], length])
    return chunks


# ---------------------------------------------------------------------------
# N1  commSignal.offsetFreq  (comm.py:63-78)
# ---------------------------------------------------------------------------
def nco(x_c64, freq_offset, samp_rate, start_index=0):
    """x * exp(-j 2 pi f (n0+n)/fs); phase in float64, the in-place ``*=`` on a
    complex64 array rounds the product once to complex64 (comm.py:77).
    ``samp_rate`` is the int-forced rate (comm.py:34)."""
    x = np.array(x_c64, dtype=np.complex64)
    n = np.arange(start_index, start_index + len(x))
    x *= np.exp(-1.0j * 2.0 * np.pi * freq_offset * n / int(samp_rate))
    return x


# ---------------------------------------------------------------------------
# SciPy restatements: lfilter_zi / lfilter / lfiltic / filtfilt
# ---------------------------------------------------------------------------
def lfilter_zi(b, a=(1.0,)):
    """Steady-state step-response state of the transposed direct form II filter
    (scipy.signal.lfilter_zi).  For an FIR (a=[1]) it reduces to
    zi[i] = sum_{k>i} b[k]  == delay line filled with ones (SURVEY.md quirk Q1,
    filters.py:45)."""
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    if a[0] != 1.0:
        b = b / a[0]
        a = a / a[0]
    n = max(len(a), len(b))
    a = np.r_[a, np.zeros(n - len(a))]
    b = np.r_[b, np.zeros(n - len(b))]
    if n == 1:
        return np.zeros(0)
    comp = np.zeros((n - 1, n - 1))
    comp[0, :] = -a[1:]
    if n > 2:
        comp[1:, :-1] = np.eye(n - 2)
    IminusA = np.eye(n - 1) - comp.T
    B = b[1:] - a[1:] * b[0]
    return np.linalg.solve(IminusA, B)


def lfilter_fir(b, x, zi=None):
    """scipy.signal.lfilter(b, [1], x, zi=zi) for an FIR: y = (b*x)[:L] with the
    transposed-form state added to the head, final state = tail of the full
    convolution (+ leftover initial state when L < ntaps-1).
    Returns (y, zf) if zi is given else y.  Output dtype is float64/complex128
    (SciPy upcasts)."""
    b = np.asarray(b, dtype=np.float64)
    x = np.asarray(x)
    cplx = np.iscomplexobj(x) or (zi is not None and np.iscomplexobj(zi))
    dt = np.complex128 if cplx else np.float64
    x = x.astype(dt)
    L = len(x)
    K = len(b)
    if L == 0:
        full = np.zeros(K - 1, dtype=dt)
    else:
        full = np.convolve(x, b.astype(dt))               # length L+K-1
    if zi is None:
        return full[:L].copy()
    zi = np.asarray(zi).astype(dt)
    zpad = np.zeros(L + K - 1, dtype=dt)
    zpad[:K - 1] = zi
    if L == 0:
        return np.zeros(0, dtype=dt), zi.copy()
    tot = full + zpad
    return tot[:L].copy(), tot[L:L + K - 1].copy()


def lfilter_df2t(b, a, x, zi=None):
    """General scipy.signal.lfilter: transposed direct form II recurrence
    (IIR; used by filters.butter, filters.py:263-273).  Pure-Python loop: small
    inputs only."""
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    b = b / a[0]
    a = a / a[0]
    n = max(len(a), len(b))
    a = np.r_[a, np.zeros(n - len(a))]
    b = np.r_[b, np.zeros(n - len(b))]
    x = np.asarray(x)
    cplx = np.iscomplexobj(x) or (zi is not None and np.iscomplexobj(zi))
    dt = np.complex128 if cplx else np.float64
    z = np.zeros(n - 1, dtype=dt) if zi is None else np.array(zi, dtype=dt)
    y = np.zeros(len(x), dtype=dt)
    for i in range(len(x)):
        xi = x[i]
        yi = b[0] * xi + (z[0] if n > 1 else 0.0)
        for k in range(n - 2):
            z[k] = z[k + 1] + b[k + 1] * xi - a[k + 1] * yi
        if n > 1:
            z[n - 2] = b[n - 1] * xi - a[n - 1] * yi
        y[i] = yi
    if zi is None:
        return y
    return y, z


def lfilter(b, a, x, zi=None):
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    if len(a) == 1:
        bb = np.asarray(b, dtype=np.float64) / a[0]
        return lfilter_fir(bb, x, zi)
    return lfilter_df2t(b, a, x, zi)


def lfiltic_fir(b, past_inputs):
    """scipy.signal.lfiltic(b, [1], y, x) for an FIR.  The reference calls
    ``lfiltic(b, a, x_chunk, initOut)`` (filters.py:66-67), i.e. the chunk lands in
    the unused ``y`` slot (a=[1] has no feedback) and ``initOut`` is taken as the
    past *inputs*, most recent first, zero-padded to ntaps-1:
        zi[m] = sum_k b[m+1+k] * x[k]."""
    b = np.asarray(b, dtype=np.float64)
    M = len(b) - 1
    x = np.zeros(M)
    p = np.asarray(past_inputs, dtype=np.float64).ravel()[:M]
    x[:len(p)] = p
    zi = np.zeros(M)
    for m in range(M):
        zi[m] = np.sum(b[m + 1:] * x[:M - m])
    return zi


def odd_ext(x, n):
    left_end = x[0]
    left_ext = x[n:0:-1]
    right_end = x[-1]
    right_ext = x[-2:-(n + 2):-1]
    return np.concatenate((2 * left_end - left_ext, x, 2 * right_end - right_ext))


def filtfilt(b, a, x):
    """scipy.signal.filtfilt(b, a, x) defaults: padtype='odd',
    padlen=3*max(len(a),len(b)), method='pad' (filters.py:72-73)."""
    b = np.atleast_1d(np.asarray(b, dtype=np.float64))
    a = np.atleast_1d(np.asarray(a, dtype=np.float64))
    x = np.asarray(x)
    ntaps = max(len(a), len(b))
    edge = ntaps * 3
    if len(x) <= edge:
        raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % edge)
    ext = odd_ext(x, edge)
    zi = lfilter_zi(b, a)
    y, _ = lfilter(b, a, ext, zi=zi * ext[0])
    y0 = y[-1]
    y, _ = lfilter(b, a, y[::-1], zi=zi * y0)
    y = y[::-1]
    return y[edge:-edge].copy()


# ---------------------------------------------------------------------------
# F1/F2/F3  filters.filter  (filters.py:21-75)
# ---------------------------------------------------------------------------
class FilterState:
    """State machine of filters.filter.__init__/applyOn (filters.py:21-75)."""

    def __init__(self, b, a=(1,), storeState=True, zeroPhase=False, initOut=None):
        self.b = np.asarray(b, dtype=np.float64)
        self.a = np.asarray(a, dtype=np.float64)
        self.storeState = storeState
        self.zeroPhase = zeroPhase
        self.initOut = initOut
        if self.storeState and self.zeroPhase:             # filters.py:38-39
            self.storeState = False
        if (self.initOut is not None) and self.zeroPhase:  # filters.py:41-42
            self.initOut = None
        self.zi = None
        if self.storeState:                                # filters.py:44-45 (unscaled zi: Q1)
            self.zi = lfilter_zi(self.b, self.a)
        if self.initOut is not None:                       # filters.py:47-48
            self.zi = None

    def applyOn(self, x):
        if self.storeState:
            if self.zi is None:                            # filters.py:66-67
                self.zi = lfiltic_fir(self.b, self.initOut)
            y, self.zi = lfilter(self.b, self.a, x, zi=self.zi)   # filters.py:69
            return y
        if self.zeroPhase:
            return filtfilt(self.b, self.a, x)             # filters.py:73
        return lfilter(self.b, self.a, x)                  # filters.py:75


def fir_history_form(b, x, hist):
    """Same FIR written the way the GPU computes it: explicit history of the last
    ntaps-1 inputs (oldest first).  history == ones reproduces Q1.
    Returns (y, new_hist).  Equivalent to lfilter_fir with the matching zi."""
    b = np.asarray(b, dtype=np.float64)
    K = len(b)
    xx = np.concatenate([np.asarray(hist), np.asarray(x)])
    dt = np.complex128 if np.iscomplexobj(xx) else np.float64
    full = np.convolve(xx.astype(dt), b.astype(dt))
    y = full[K - 1:K - 1 + len(x)]
    return y.copy(), xx[len(xx) - (K - 1):].copy()


# window designs used by the filter classes (filters.py:139,199,226; SciPy
# windows.* with sym=True -- closed forms)
def win_hamming(n):
    if n == 1:
        return np.ones(1)
    k = np.arange(n)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * k / (n - 1))


def win_blackmanharris(n):
    if n == 1:
        return np.ones(1)
    k = np.arange(n)
    f = 2.0 * np.pi * k / (n - 1)
    return 0.35875 - 0.48829 * np.cos(f) + 0.14128 * np.cos(2 * f) - 0.01168 * np.cos(3 * f)


def win_gaussian(n, sigma):
    k = np.arange(n) - (n - 1.0) / 2.0
    return np.exp(-k ** 2 / (2.0 * sigma * sigma))


# ---------------------------------------------------------------------------
# R1  commSignal.bwLim non-strict  (comm.py:107-108,118-130)
# ---------------------------------------------------------------------------
def decimate_carry(x, samp_rate, t_samp_rate, offset=0):
    """Returns (x[offset::M], new_rate, next_offset, M).  ``offset`` is the chunker
    variable 'bwlim'+uniq (comm.py:123-125)."""
    if samp_rate < t_samp_rate:
        raise ValueError("The target sampling rate must be less than current sampling rate")
    M = int(samp_rate / t_samp_rate)
    L = len(x)
    next_off = (M - (L - offset) % M) % M
    return x[offset::M], int(samp_rate / M), next_off, M


# ---------------------------------------------------------------------------
# D1  demod_fm.demod  (demod_fm.py:29-51)
# ---------------------------------------------------------------------------
def fm_demod(sig, last=None, store_state=True):
    """Returns (angles, new_last).  First call -> L-1 outputs, later calls L (Q3)."""
    sig = np.asarray(sig)
    d = sig[1:] * np.conj(sig[:-1])
    if store_state:
        if last is None:
            return np.angle(d), sig[-1]
        corr = np.array([sig[0] * np.conj(last)])
        return np.angle(np.concatenate([corr, d])), sig[-1]
    return np.angle(d), None


# ---------------------------------------------------------------------------
# R2  commSignal.bwLim strict -> scipy.signal.resample  (comm.py:110-116)
# ---------------------------------------------------------------------------
# ---------------------------------------------------------------------------
# Polyphase rational resampler (BASELINE north_star / config 3: "polyphase resample to 11.025 kS/s").
# The reference has NO counterpart (its only resampler is the FFT one below, comm.py:110-116), so this stage is
# build-defined; it follows SciPy's published scipy.signal.resample_poly (v1.15: firwin low-pass with a
# Kaiser(5.0) window of half length 10*max(up, down), scaled by `up`, zero-padded in front so that output j sits
# at input time j*down/up, upfirdn, excess removed) and is pinned against that routine in the tests.
# ---------------------------------------------------------------------------
def kaiser_window(M, beta):
    n = np.arange(M)
    alpha = (M - 1) / 2.0
    return np.i0(beta * np.sqrt(np.clip(1.0 - ((n - alpha) / alpha) ** 2, 0.0, 1.0))) / np.i0(beta)


def firwin_lowpass(numtaps, cutoff, beta=5.0):
    """scipy.signal.firwin(numtaps, cutoff, window=('kaiser', beta)) for one low-pass band (cutoff relative to Nyquist)"""
    alpha = 0.5 * (numtaps - 1)
    m = np.arange(numtaps) - alpha
    h = cutoff * np.sinc(cutoff * m) * kaiser_window(numtaps, beta)
    return h / np.sum(h)                                   # unity gain at DC (scale=True)


def resample_poly_design(up, down):
    """(up, down reduced, padded taps hp, n_pre_remove): y[j] = sum_k hp[k] xu[(j + n_pre_remove) down - k]"""
    g = math.gcd(int(up), int(down))
    up, down = int(up) // g, int(down) // g
    half_len = 10 * max(up, down)
    h = firwin_lowpass(2 * half_len + 1, 1.0 / max(up, down)) * up
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    return up, down, np.concatenate((np.zeros(n_pre_pad), h)), n_pre_remove


def resample_poly(x, up, down):
    """scipy.signal.resample_poly(x, up, down) for a real 1-D signal (padtype 'constant', zeros)"""
    x = np.asarray(x, dtype=np.float64)
    up, down, hp, npr = resample_poly_design(up, down)
    if up == down == 1:
        return x.copy()
    n_in = len(x)
    n_out = -(-n_in * up // down)
    # only the taps k = k0 + q up meet a non-zero sample of the zero-stuffed input: one polyphase branch per output,
    # walked for all outputs at once (q-th tap of every branch per step)
    t = (np.arange(n_out, dtype=np.int64) + npr) * down
    k0 = t % up
    i0 = (t - k0) // up
    y = np.zeros(n_out)
    for q in range(-(-len(hp) // up)):
        k = k0 + q * up
        i = i0 - q
        ok = (k < len(hp)) & (i >= 0) & (i < n_in)
        if not ok.any():
            continue
        y[ok] += hp[k[ok]] * x[i[ok]]
    return y


class PolyResampler:
    """The same resampler as a stream: chunks in, the outputs that have become computable out; flush() emits the
    tail.  Concatenated, the outputs equal resample_poly of the concatenated input (the build's chunked form)."""

    def __init__(self, up, down):
        self.up, self.down, self.hp, self.npr = resample_poly_design(up, down)
        self.buf = np.zeros(0)
        self.n_in = 0
        self.j_next = 0

    def _emit(self, j_last):
        out = np.zeros(max(0, j_last - self.j_next + 1))
        for j in range(self.j_next, j_last + 1):
            t = (j + self.npr) * self.down
            k0 = t % self.up
            i0 = (t - k0) // self.up
            k = np.arange(k0, len(self.hp), self.up)
            i = i0 - np.arange(len(k))
            ok = (i >= 0) & (i < self.n_in)
            out[j - self.j_next] = np.sum(self.hp[k[ok]] * self.buf[i[ok]])
        self.j_next = max(self.j_next, j_last + 1)
        return out

    def applyOn(self, x):
        x = np.asarray(x, dtype=np.float64)
        self.buf = np.concatenate((self.buf, x))
        self.n_in += len(x)
        if self.n_in == 0:
            return np.zeros(0)
        return self._emit((self.n_in * self.up - 1) // self.down - self.npr)   # every input an output needs has arrived

    def flush(self):
        return self._emit(-(-self.n_in * self.up // self.down) - 1)


def resample_fft(x, num):
    """scipy.signal.resample(x, num): Fourier-domain resampling of the whole chunk."""
    x = np.asarray(x)
    Nx = len(x)
    real = np.isrealobj(x)
    X = np.fft.rfft(x) if real else np.fft.fft(x)
    Y = np.zeros(num // 2 + 1 if real else num, dtype=X.dtype)
    N = min(num, Nx)
    nyq = N // 2 + 1
    Y[:nyq] = X[:nyq]
    if not real and N > 2:
        Y[nyq - N:] = X[nyq - N:]
    if N % 2 == 0:
        if num < Nx:
            if real:
                Y[N // 2] *= 2.0
            else:
                Y[-N // 2] += X[-N // 2]
        elif Nx < num:
            Y[N // 2] *= 0.5
            if not real:
                Y[num - N // 2] = Y[N // 2]
    y = np.fft.irfft(Y, num) if real else np.fft.ifft(Y)
    y *= float(num) / float(Nx)
    return y


def bwlim_strict(x, samp_rate, t_samp_rate):
    """comm.py:110-116: returns (resampled, new_rate)."""
    if samp_rate < t_samp_rate:
        raise ValueError("The target sampling rate must be less than current sampling rate")
    return resample_fft(x, int(t_samp_rate * len(x) / samp_rate)), t_samp_rate


# ---------------------------------------------------------------------------
# A1  demod_am.demod = abs(hilbert(x))  (demod_am.py:18-29)
# ---------------------------------------------------------------------------
def hilbert(x):
    x = np.asarray(x, dtype=np.float64)
    N = len(x)
    Xf = np.fft.fft(x)
    h = np.zeros(N)
    if N % 2 == 0:
        h[0] = h[N // 2] = 1
        h[1:N // 2] = 2
    else:
        h[0] = 1
        h[1:(N + 1) // 2] = 2
    return np.fft.ifft(Xf * h)


def am_demod(x):
    return np.abs(hilbert(x))


def am_demod_blocks(x, block=60000 * 4):
    """decode_noaa.__getAM (decode_noaa.py:631-657): fixed 240 000-sample blocks,
    block list by chunker rule, no overlap."""
    out = [am_demod(x[a:b]) for a, b in chunk_list(len(x), block)]
    return np.concatenate(out) if out else np.zeros(0)


# ---------------------------------------------------------------------------
# X1  decode_noaa.__correlate  (decode_noaa.py:659-675)
# ---------------------------------------------------------------------------
def correlate_same(a, v):
    """scipy.signal.correlate(a, v, mode='same') for real 1-D, len(v) <= len(a):
    full cross-correlation centred on the first input."""
    a = np.asarray(a, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    full = _fft_convolve_full(a, v[::-1])
    start = (len(full) - len(a)) // 2
    return full[start:start + len(a)]


def _fft_convolve_full(a, b):
    n = len(a) + len(b) - 1
    if min(len(a), len(b)) < 64:
        return np.convolve(a, b)
    nf = 1 << int(math.ceil(math.log2(n)))
    return np.fft.irfft(np.fft.rfft(a, nf) * np.fft.rfft(b, nf), nf)[:n]


def window_energy_same(h, m):
    """np.convolve(h*h, [1]*m, mode='same') (decode_noaa.py:672) as an O(N) prefix
    sum: out[i] = sum_{j} h2[i + (m-1)//2 ... ] with numpy's 'same' centring."""
    h2 = np.asarray(h, dtype=np.float64) ** 2
    n = len(h2)
    c = np.concatenate([[0.0], np.cumsum(h2)])
    # full[k] = sum_{j=max(0,k-m+1)}^{min(k,n-1)} h2[j];  same = full[(m-1)//2 : (m-1)//2+n]
    k = np.arange(n) + (m - 1) // 2
    lo = np.maximum(0, k - m + 1)
    hi = np.minimum(k, n - 1)
    return c[hi + 1] - c[lo]


def xcorr_norm(haystack, needle, exact_energy=False):
    """decode_noaa.__correlate (decode_noaa.py:671-673)."""
    haystack = np.asarray(haystack, dtype=np.float64)
    needle = np.asarray(needle, dtype=np.float64)
    cor = correlate_same(haystack, needle)
    if exact_energy:
        sums = np.convolve(haystack * haystack, [1] * len(needle), mode="same")
    else:
        sums = window_energy_same(haystack, len(needle))
    return cor / (sums * np.sum(needle * needle)) ** 0.5


# ---------------------------------------------------------------------------
# X2  decode_noaa.__correlateAndFindPeaks  (decode_noaa.py:677-767)
# ---------------------------------------------------------------------------
def sync_needle(sync_bits, samp_rate, pos=True):
    rep = round(samp_rate * NOAA_T)                        # decode_noaa.py:689
    if pos:
        return ((np.repeat(sync_bits, rep) * 233) + 11) / 255   # :691
    return np.repeat(sync_bits, rep) - 0.5                 # :693


def find_peaks(cor, samp_rate, needle_len):
    """Peak pick of decode_noaa.py:713-751.  Returns int64 indices (start of sync)."""
    cor = np.asarray(cor, dtype=np.float64)
    K = int(2 * (len(cor) / samp_rate)) + 2                # :714
    maxk = np.argpartition(cor, -1 * K)[-1 * K:]           # :717
    avgpk = np.sum(cor[maxk]) / K                          # :720
    avgpk -= NOAA_PEAKHEIGHTWIGGLE * (avgpk - (np.sum(cor[np.argpartition(cor, K)[:K]]) / K))   # :723
    possible = np.sort(np.argwhere(cor > avgpk).ravel())   # :726
    min_dist = NOAA_MINPEAKDIST * samp_rate                # :729
    peaks = []
    cur_max = None
    cur_idx = None
    for i in possible:                                     # :736
        if cur_idx is not None and (i - cur_idx) >= min_dist:
            peaks.append(cur_idx)
            cur_max = None
            cur_idx = None
        if cur_max is None or cur_max < cor[i]:            # strict <, first max wins (:742)
            cur_max = cor[i]
            cur_idx = i
    peaks.append(cur_idx)                                  # :746
    peaks = [int(i) - int(needle_len / 2) for i in peaks]  # :749
    return np.sort(np.array(peaks, dtype=np.int64).ravel())


def correlate_and_find_peaks(sig, samp_rate, sync_bits, use_filter_taps=None, extra=False):
    """decode_noaa.py:677-767 with useNormCorrelate=True, usePosNeedle=True.
    ``use_filter_taps``: taps of the zero-phase pre-filter (hamming(492) default
    argument, decode_noaa.py:677) or None."""
    needle = sync_needle(sync_bits, samp_rate)
    s = np.asarray(sig, dtype=np.float64)
    hay = filtfilt(use_filter_taps, [1.0], s) if use_filter_taps is not None else s
    cor = xcorr_norm(hay, needle)
    peaks = find_peaks(cor, samp_rate, len(needle))
    if not extra:
        return peaks
    n = len(needle)
    heights, tsync = [], []
    for i in peaks:                                        # :754-762
        if i + 2 * n < len(s):
            tsync.append(float(np.average(s[i + n:i + 2 * n])))
        else:
            tsync.append(None)
        heights.append(float(cor[i + int(n / 2)]))
    return peaks, heights, tsync


# ---------------------------------------------------------------------------
# P  pipeline drivers (stage order from the callers; not re-implementations of them)
# ---------------------------------------------------------------------------
class ChunkState(dict):
    """chunker.get/set dictionary (chunker.py:54-84)."""


def audio_chain(read, length, fs, freq_offset, taps, bw, audio_rate=None, strict=False,
                chunk_size=PROC_CHUNKSIZE, use_nco=True):
    """decode_noaa.__audio / decode_fm.getAudio / tutorial 3 chunk loop
    (decode_noaa.py:600-629, decode_fm.py:42-72): NCO -> FIR(state) -> bwLim ->
    FM(state) -> bwLim(audio_rate, strict).  ``read(a,b)`` returns complex64.
    Returns (audio float64, rate)."""
    filt = FilterState(taps)
    fm_last = None
    nco_index = 0
    dec_off = 0
    dec_off2 = 0
    out = []
    rate = None
    for a, b in chunk_list(length, chunk_size):
        x = np.array(read(a, b), dtype=np.complex64)
        if use_nco:
            x = nco(x, freq_offset, fs, nco_index)
            nco_index += len(x)
        y = filt.applyOn(x)
        y, r1, dec_off, _ = decimate_carry(y, int(fs), bw, dec_off)
        ang, fm_last = fm_demod(y, fm_last)
        rate = r1
        if audio_rate is not None:
            if strict:
                ang, rate = bwlim_strict(ang, r1, audio_rate)
            else:
                ang, rate, dec_off2, _ = decimate_carry(ang, r1, audio_rate, dec_off2)
        out.append(ang)
    return np.concatenate(out), rate


def crude_sync(audio, rate):
    """decode_noaa.getCrudeSync after __audio (decode_noaa.py:781-804).
    Returns (syncA, syncB, useful)."""
    am = am_demod_blocks(audio)
    sa = correlate_and_find_peaks(am, rate, NOAA_SYNCA)
    sb = correlate_and_find_peaks(am, rate, NOAA_SYNCB)
    useful = 0

    def _min_dev(s):
        d = np.abs(np.diff(s) - (rate * 0.5))
        if len(d) - NOAA_DETECTCONSSYNCSNUM + 1 <= 0:
            return np.inf
        return np.min([np.max(d[i:i + NOAA_DETECTCONSSYNCSNUM])
                       for i in range(len(d) - NOAA_DETECTCONSSYNCSNUM + 1)])
    if _min_dev(sa) < NOAA_DETECTMAXCHANGE or _min_dev(sb) < NOAA_DETECTMAXCHANGE:
        useful = 1
    return sa, sb, useful


def accurate_sync_window(x_c64, fs, freq_offset, sync_bits):
    """One window of decode_noaa.getAccurateSync (decode_noaa.py:852-853):
    NCO(no carry) -> BH151 filtfilt -> FM(fresh) -> AM -> hamming(492) filtfilt ->
    normalised correlation -> first peak.  Returns (index_in_window, height, timesync)."""
    x = nco(x_c64, freq_offset, fs, 0)
    y = filtfilt(win_blackmanharris(151), [1.0], x)
    ang, _ = fm_demod(y, None)
    am = am_demod(ang)
    pk, ht, ts = correlate_and_find_peaks(am, int(fs), sync_bits,
                                          use_filter_taps=win_hamming(492), extra=True)
    return int(pk[0]), ht[0], ts[0]


# ---------------------------------------------------------------------------
# synthetic inputs; values on the source.read grid
# ---------------------------------------------------------------------------
def synth_iq_noise(n, seed):
    """Input A: I,Q iid uniform integers 0..255 minus 127.5 (source.py:117-118 grid)."""
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 256, size=(n, 2), dtype=np.uint8)
    return raw


def synth_iq_fm(n, fs, seed, f_carrier=25e3, f_mod=1e3, dev=5.0, amp=60.0, sigma=4.0, start=0):
    """Input B: FM tone + noise, rounded to the u8 grid."""
    rng = np.random.default_rng(seed)
    t = (np.arange(start, start + n)) / fs
    s = amp * np.exp(1j * (2 * np.pi * f_carrier * t + dev * np.sin(2 * np.pi * f_mod * t)))
    s = s + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    raw = np.empty((n, 2), dtype=np.uint8)
    raw[:, 0] = np.clip(np.round(s.real + 127.5), 0, 255).astype(np.uint8)
    raw[:, 1] = np.clip(np.round(s.imag + 127.5), 0, 255).astype(np.uint8)
    return raw


def synth_apt_iq(duration_s, fs=2048000, seed=1, f_offset=30000.0, dev=17000.0,
                 amp=60.0, sigma=4.0):
    """Synthetic NOAA-APT-shaped IQ: 2 lines/s x 2080 words at
    4160 words/s; sync A at words 0-39, sync B at 1040-1079 mapped (bit*233+11)/255;
    AM on a 2400 Hz subcarrier; FM (dev Hz) placed at +f_offset; u8 grid."""
    rng = np.random.default_rng(seed)
    n = int(duration_s * fs)
    nwords = int(math.ceil(duration_s * 4160)) + 1
    words = rng.uniform(0.2, 0.8, size=nwords)
    for line_start in range(0, nwords, 2080):
        for k in range(40):
            if line_start + k < nwords:
                words[line_start + k] = (NOAA_SYNCA[k] * 233 + 11) / 255.0
            if line_start + 1040 + k < nwords:
                words[line_start + 1040 + k] = (NOAA_SYNCB[k] * 233 + 11) / 255.0
    out = np.empty((n, 2), dtype=np.uint8)
    blk = 1 << 20
    phase = 0.0
    for s0 in range(0, n, blk):
        s1 = min(n, s0 + blk)
        idx = np.arange(s0, s1)
        t = idx / fs
        env = words[np.minimum((idx * 4160) // fs, nwords - 1)]
        audio = env * np.sin(2 * np.pi * 2400.0 * t)
        ph = phase + 2 * np.pi * dev * np.cumsum(audio) / fs
        phase = ph[-1]
        s = amp * np.exp(1j * (2 * np.pi * f_offset * t + ph))
        s = s + sigma * (rng.standard_normal(s1 - s0) + 1j * rng.standard_normal(s1 - s0))
        out[s0:s1, 0] = np.clip(np.round(s.real + 127.5), 0, 255).astype(np.uint8)
        out[s0:s1, 1] = np.clip(np.round(s.imag + 127.5), 0, 255).astype(np.uint8)
    return out


def grid_c64(raw_u8):
    """u8[N,2] -> complex64 on the source.read grid."""
    return read_iq_u8(raw_u8, 0, len(raw_u8))


# ------------------------------------------------------------------ AFSK1200 correlators
# decode_afsk1200.py:99-158 restated: the quadrature correlators for the mark (1200 Hz) and
# space (2200 Hz) tones over one baud, their power difference, and the bit-edge detector.
def afsk_tables(bw, baud=1200, mark=1200, space=2200):
    """decode_afsk1200.py:99-123: buffer_size = round(bw/baud) samples of cos/sin at both tones;
    returns (tables[4, buffer_size] = mark_i, mark_q, space_i, space_q, samples_per_baud)."""
    bs = int(np.round(bw / baud))
    spb = bw // baud
    tb = np.zeros((4, bs))
    for i in range(bs):
        ma = (i * 1.0 / bw) / (1 / mark) * 2 * np.pi
        sa = (i * 1.0 / bw) / (1 / space) * 2 * np.pi
        tb[0, i], tb[1, i], tb[2, i], tb[3, i] = np.cos(ma), np.sin(ma), np.cos(sa), np.sin(sa)
    return tb, int(spb)


def afsk_binary_filter(sig, tables):
    """decode_afsk1200.py:126-141.  out[s] = mi^2 + mq^2 - si^2 - sq^2 with the four sums taken
    in the reference's order (sub = 0 .. buffer_size-1, product then add, float64); the last
    buffer_size entries stay 0 (the loop stops at len - buffer_size)."""
    sig = np.asarray(sig, dtype=np.float64)
    bs = tables.shape[1]
    n = len(sig) - bs
    out = np.zeros(len(sig))
    if n <= 0:
        return out
    acc = np.zeros((4, n))
    for sub in range(bs):
        seg = sig[sub:sub + n]
        for c in range(4):
            acc[c] = acc[c] + seg * tables[c, sub]
    out[:n] = acc[0] ** 2 + acc[1] ** 2 - acc[2] ** 2 - acc[3] ** 2
    return out


def afsk_edges(binary_filter, spb):
    """decode_afsk1200.py:147-156: correlate(sign(bf), [-1]*(spb//2) + [1]*(spb - spb//2), 'same') / spb"""
    kernel = np.where(np.arange(spb) < spb // 2, -1.0, 1.0)
    return np.correlate(np.sign(binary_filter), kernel, mode="same") / spb


def synth_afsk_iq(n_bits, fs, seed, bw=22050, baud=1200, mark=1200, space=2200, dev=3000.0, amp=60.0, sigma=2.0, f_carrier=0.0):
    """FM-modulated AFSK1200 (random NRZI bits) as u8 IQ at `fs` (a multiple of bw); f_carrier: where the signal sits in the recording
    (config 1: 10 kHz above the centre frequency the file name carries)."""
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, n_bits)
    spb_fs = fs / baud
    n = int(n_bits * spb_fs)
    t = np.arange(n)
    tone = np.where(bits[np.minimum((t / spb_fs).astype(np.int64), n_bits - 1)] == 1, mark, space).astype(np.float64)
    audio_phase = 2 * np.pi * np.cumsum(tone) / fs
    audio = np.cos(audio_phase)
    rf_phase = 2 * np.pi * dev * np.cumsum(audio) / fs
    if f_carrier:
        rf_phase = rf_phase + 2 * np.pi * ((f_carrier / fs * t) % 1.0)
    s = amp * np.exp(1j * rf_phase) + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    raw = np.empty((n, 2), dtype=np.uint8)
    raw[:, 0] = np.clip(np.round(s.real + 127.5), 0, 255).astype(np.uint8)
    raw[:, 1] = np.clip(np.round(s.imag + 127.5), 0, 255).astype(np.uint8)
    return raw
