"""
Thin wrappers that launch the C-ABI entry points on device arrays (internal).
"""
import ctypes as C

import numpy as np

from . import _hip
from ._hip import DevArray, check, lib

_C64 = np.dtype(np.complex64)
_F32 = np.dtype(np.float32)
_F64 = np.dtype(np.float64)
_C128 = np.dtype(np.complex128)

# test/bench hook: force the f32 direct-form kernels even where the MFMA path applies
FORCE_DIRECT = False


def nco(x, cyc, start):
    """N1 (comm.py:63-78)"""
    if x.dtype != _C64:
        raise TypeError("offsetFreq needs complex IQ samples")
    out = DevArray(x.n, _C64)
    check(lib().dd_nco_c64(x.ptr, out.ptr, x.n, cyc, start, None), "dd_nco_c64")
    return out


def decimate(x, m, off):
    """R1 (comm.py:118-130)"""
    n_out = len(range(off, x.n, m))
    out = DevArray(n_out, x.dtype)
    no = C.c_int64(0)
    check(lib().dd_decimate(x.ptr, out.ptr, x.n, m, off, x.dtype.itemsize, C.byref(no), None), "dd_decimate")
    assert no.value == n_out
    return out


def fused(x, filt, nco_op, decim, fm):
    """One kernel for [offsetFreq] -> filter -> [bwLim] -> [demod_fm.demod]."""
    m, off = decim
    fir_h = filt._handle()
    carry = filt._prepare_call()
    fm_h = None
    if fm is not None:
        fm_h = fm._handle()
        fm._prepare_call()
    kept = len(range(off, x.n, m))
    if fm is not None:
        n_expect = max(0, kept - (0 if fm._dev_has_last() else 1))
        out = DevArray(n_expect, _F32)
    else:
        n_expect = kept
        out = DevArray(n_expect, _C64)
    no = C.c_int64(0)
    flags = _hip.DD_CHAIN_FORCE_DIRECT if FORCE_DIRECT else 0
    if x.dtype == _hip.IQ8:
        flags |= _hip.DD_CHAIN_U8_INPUT
    if getattr(filt, "tight", False):
        flags |= _hip.DD_CHAIN_TIGHT
    check(lib().dd_fused_process(fir_h, fm_h, x.ptr, out.ptr, x.n,
                                 1 if nco_op is not None else 0,
                                 nco_op[1] if nco_op is not None else 0,
                                 nco_op[2] if nco_op is not None else 0,
                                 m, off, flags, 1 if carry else 0, C.byref(no), None), "dd_fused_process")
    assert no.value == n_expect, (no.value, n_expect)
    if fm is not None:
        fm._after_call()
    return out


def resample_fft(x, num):
    """R2 (comm.py:110-116 -> scipy.signal.resample), float64 real"""
    if x.dtype == _F32:
        from .comm import _convert
        x = _convert(x, _F64)
    if x.dtype != _F64:
        raise NotImplementedError("strict bwLim (FFT resample) is implemented for real signals")
    out = DevArray(num, _F64)
    check(lib().dd_resample_fft_f64(x.ptr, out.ptr, x.n, num, None), "dd_resample_fft_f64")
    return out


def resample_fft_chunks(x, in_offsets, lengths, nums):
    """R2 for a chunk list in one call (dd_resample_fft_chunks): chunk i = x[in_offsets[i] : in_offsets[i] + lengths[i]]
    (float32 or float64 device array) resampled to nums[i] samples; returns the concatenated float64 device array and
    the output offsets"""
    import ctypes as C
    if x.dtype not in (_F32, _F64):
        raise NotImplementedError("strict bwLim (FFT resample) is implemented for real signals")
    k = len(lengths)
    out_off = [0] * k
    for i in range(1, k):
        out_off[i] = out_off[i - 1] + int(nums[i - 1])
    total = (out_off[-1] + int(nums[-1])) if k else 0
    out = DevArray(max(total, 1), _F64)
    A = C.c_int64 * max(k, 1)
    check(lib().dd_resample_fft_chunks(x.ptr, 1 if x.dtype == _F32 else 0, A(*[int(v) for v in in_offsets]), A(*[int(v) for v in lengths]), out.ptr,
                                       A(*out_off), A(*[int(v) for v in nums]), k, None), "dd_resample_fft_chunks")
    return out.view(0, total), out_off


def am_envelope(x, block=None):
    """A1 (demod_am.py:18-29)"""
    if x.dtype == _F32:
        from .comm import _convert
        x = _convert(x, _F64)
    if x.dtype != _F64:
        raise TypeError("demod_am expects a real signal")
    out = DevArray(x.n, _F64)
    check(lib().dd_am_envelope_f64(x.ptr, out.ptr, x.n, block if block else max(1, x.n), None), "dd_am_envelope_f64")
    return out


def xcorr_norm(h, needle):
    """X1 (decode_noaa.py:659-675)"""
    needle = np.ascontiguousarray(needle, dtype=np.float64)
    out = DevArray(h.n, _F64)
    check(lib().dd_xcorr_norm_f64(h.ptr, h.n, needle.ctypes.data_as(C.POINTER(C.c_double)), len(needle),
                                  out.ptr, None), "dd_xcorr_norm_f64")
    return out


def find_peaks(cor, samp_rate, needle_len, max_peaks=None):
    """X2 peak pick (decode_noaa.py:713-751) -> sorted int64 indices"""
    if max_peaks is None:
        # peak groups are at least 0.45 s apart (decode_noaa.py:729,737): cor.n / (0.45 fs) + 1 at most
        max_peaks = int(cor.n / (0.45 * samp_rate)) + 2 + 64
    buf = (C.c_int64 * max_peaks)()
    n = C.c_int(0)
    check(lib().dd_find_peaks_f64(cor.ptr, cor.n, float(samp_rate), int(needle_len), buf, max_peaks,
                                  C.byref(n), None), "dd_find_peaks_f64")
    return np.array(buf[:n.value], dtype=np.int64)


def noaa_prepare(n_audio, block=60000 * 4, window=0):
    """what crude_tail will need for n_audio samples and the accurate-sync windows for `window` IQ samples each, built ahead
    (dd_noaa_prepare); errors are left to the calls themselves"""
    try:
        lib().dd_noaa_prepare(int(n_audio), int(block), int(window), None)
    except Exception:
        pass


def crude_tail(audio, samp_rate, needles, block=60000 * 4, want_env=False):
    """getCrudeSync's audio-rate tail in one device call (decode_noaa.py:781-790): envelope in `block`-sample blocks, then
    normalised correlation + peak pick for every needle (same length).  Returns ([peaks per needle], envelope DevArray or
    None), or None when the fused entry declines (dd_noaa_crude_tail: DD_ERR_UNSUPPORTED) and the caller must go stage by stage."""
    from . import _hip
    nd = np.ascontiguousarray(np.stack([np.asarray(v, dtype=np.float64) for v in needles]))
    nn, m = nd.shape
    if audio.dtype not in (_F32, _F64) or nn > 2:
        return None
    max_peaks = int(audio.n / (0.45 * samp_rate)) + 2 + 64
    buf = (C.c_int64 * (max_peaks * nn))()
    cnt = (C.c_int * nn)()
    env = DevArray(audio.n, _F64) if want_env else None
    rc = lib().dd_noaa_crude_tail(audio.ptr, 1 if audio.dtype == _F32 else 0, audio.n, float(samp_rate), int(block),
                                  nd.ctypes.data_as(C.POINTER(C.c_double)), int(m), int(nn), env.ptr if env is not None else None,
                                  buf, max_peaks, cnt, None)
    if rc == _hip.DD_ERR_UNSUPPORTED:
        return None
    check(rc, "dd_noaa_crude_tail")
    return [np.array(buf[d * max_peaks:d * max_peaks + cnt[d]], dtype=np.int64) for d in range(nn)], env


def filtfilt(taps, x):
    """F2 (filters.py:72-73 -> scipy.signal.filtfilt)"""
    t = np.ascontiguousarray(taps, dtype=np.float64)
    tp = t.ctypes.data_as(C.POINTER(C.c_double))
    if x.dtype == _F32:
        from .comm import _convert
        x = _convert(x, _F64)
    out = DevArray(x.n, x.dtype)
    if x.dtype == _C64:
        check(lib().dd_filtfilt_c64(tp, len(t), x.ptr, out.ptr, x.n, None), "dd_filtfilt_c64")
    elif x.dtype == _F64:
        check(lib().dd_filtfilt_f64(tp, len(t), x.ptr, out.ptr, x.n, 0, None), "dd_filtfilt_f64")
    elif x.dtype == _C128:
        check(lib().dd_filtfilt_f64(tp, len(t), x.ptr, out.ptr, x.n, 1, None), "dd_filtfilt_f64")
    else:
        raise TypeError("unsupported dtype %s" % x.dtype)
    return out
