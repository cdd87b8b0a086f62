"""
Filters -- drop-in for the reference's directdemod/filters.py (base ``filter`` with
``applyOn``/``getA``/``getB`` and the ``rollingAverage``, ``blackmanHarris``,
``hamming``, ``gaussian``, ``butter``, ``remez`` designs; same constructor
arguments and error behaviour).

Arithmetic.  An FIR (a == [1]) is applied by the LDS-tiled HIP kernels behind
dd_fir_* / dd_fused_process; its chunk-to-chunk state is the last ntaps-1 input
samples, kept in HBM inside the C handle.  The reference seeds SciPy's ``lfilter``
with ``lfilter_zi(b, a)`` *without* scaling by the first sample (filters.py:45);
for an FIR that is exactly a delay line pre-filled with 1.0, which is how the
handle is initialised (quirk Q1).  Taps are the raw, un-normalised windows (Q2).
``zeroPhase`` is SciPy's ``filtfilt`` (odd extension 3*ntaps, forward-backward).
"""
import ctypes as C

import numpy as np

from . import _hip, _ops, constants
from ._hip import DevArray, check, lib

_C64 = np.dtype(np.complex64)
_F32 = np.dtype(np.float32)
_F64 = np.dtype(np.float64)


# Device handles (dd_fir: taps in every kernel's layout, history ping-pong, constant histories) of filter objects that have
# been garbage-collected, kept for the next filter object with the same taps: the reference's callers build their filters inside
# the call (decode_noaa.py:611-613 builds blackmanHarris(151) in every __audio), and creating / destroying a handle is a dozen
# hipMalloc / hipFree calls with device synchronisations (0.5 ms of getCrudeSync's 2.1 ms, tools/debug/noaa_hostprof.py).  A
# handle taken from here starts from the first-call state (history of ones, quirk Q1) like a new one.
_FIR_POOL = {}
_FIR_POOL_PER_KEY, _FIR_POOL_TOTAL = 4, 32


def _pool_take(key):
    lst = _FIR_POOL.get(key)
    if lst:
        return lst.pop()
    return None


def _pool_give(key, h):
    if sum(len(v) for v in _FIR_POOL.values()) >= _FIR_POOL_TOTAL:
        return False
    lst = _FIR_POOL.setdefault(key, [])
    if len(lst) >= _FIR_POOL_PER_KEY:
        return False
    lst.append(h)
    return True


def pool_clear():
    """destroy the pooled handles (tests, and before the library is unloaded)"""
    for lst in _FIR_POOL.values():
        for h in lst:
            lib().dd_fir_destroy(h)
    _FIR_POOL.clear()


class filter:
    '''
    Parent object of all filters (filters.py:15-89).
    '''

    def __init__(self, b, a, storeState=True, zeroPhase=False, initOut=None):
        '''Args:
            b (:obj:`list`): 'b' constants of filter
            a (:obj:`list`): 'a' constants of filter
            storeState (:obj:`bool`, optional): carry the filter state from call to call
            zeroPhase (:obj:`bool`, optional): forward-backward filtering, no delay
                (disables 'storeState' and 'initOut', filters.py:38-42)
            initOut (:obj:`list`, optional): initial condition (past inputs, most recent first)
        '''
        self.__storeState = storeState
        self.__zeroPhase = zeroPhase
        self.__initOut = initOut
        if self.__storeState and self.__zeroPhase:
            self.__storeState = False
        if (self.__initOut is not None) and self.__zeroPhase:
            self.__initOut = None
        self.__b = b
        self.__a = a
        self.__h = None
        self.__iir = None
        self.__seeded = self.__initOut is None      # initOut history still to be loaded?
        av = np.atleast_1d(np.asarray(a, dtype=np.float64))
        self.__isFIR = av.size == 1
        self.__taps = np.atleast_1d(np.asarray(b, dtype=np.float64)) / av[0] if self.__isFIR else None
        # not in the reference (float64 there): True keeps the running-sum kernel off for this filter (DD_CHAIN_TIGHT) -- hamming(255)
        # without decimation then takes the transform kernel, whose FM angles of a signal IN THE STOP BAND are good to 3e-4 rad
        # instead of 1e-3 (1.4 x the time).  Set it on the object: flt = filters.hamming(255); flt.tight = True
        self.tight = False

    # -- device handle ------------------------------------------------------------
    def _fusable(self):
        return self.__isFIR and not self.__zeroPhase

    def _carries(self):
        """storeState: the history moves on from call to call (what a chunk-list launch needs)"""
        return bool(self.__storeState)

    def _last_kernel(self):
        """DD_KERNEL_* of the last fused launch through this filter (tests assert the intended kernel ran)"""
        return int(lib().dd_fir_last_kernel(self._handle()))

    def _launch_count(self):
        """fused kernel launches through this filter since it was created or last reset (a chunk list in one launch counts once)"""
        return int(lib().dd_fir_launch_count(self._handle()))

    def _handle(self):
        if self.__h is None:
            _hip.require_gpu()
            t = np.ascontiguousarray(self.__taps, dtype=np.float64)
            self.__key = t.tobytes()
            p = _pool_take(self.__key)
            if p is not None:
                check(lib().dd_fir_reset(p, _hip.DD_HIST_ONES, None, None), "dd_fir_reset")       # launch-free: a constant history buffer
            else:
                p = C.c_void_p()
                check(lib().dd_fir_create(C.byref(p), t.ctypes.data_as(C.POINTER(C.c_double)), len(t)), "dd_fir_create")
            self.__h = p
        return self.__h

    def __del__(self):
        try:
            if self.__h is not None:
                if not _pool_give(self.__key, self.__h):
                    lib().dd_fir_destroy(self.__h)
                self.__h = None
            if self.__iir is not None:
                lib().dd_iir_destroy(self.__iir)
        except Exception:
            pass

    def _prepare_call(self):
        """Bring the device history to what SciPy's state would be before this call;
        returns the carry flag."""
        h = self._handle()
        if self.__storeState:
            if not self.__seeded:
                # lfiltic(b, a, x, initOut) with a=[1]: initOut are the past inputs, most
                # recent first, zero padded (filters.py:66-67)
                k1 = len(self.__taps) - 1
                past = np.zeros(k1, dtype=np.float64)
                io = np.asarray(self.__initOut, dtype=np.float64).ravel()[:k1]
                past[:len(io)] = io
                hist = np.zeros(max(1, k1), dtype=np.complex64)
                hist[:k1] = past[::-1]
                check(lib().dd_fir_reset(h, _hip.DD_HIST_GIVEN, hist.ctypes.data, None), "dd_fir_reset")
                if k1 > 0:
                    # the float64 real path takes the history as doubles (complex64 would round initOut to float32)
                    h64 = np.ascontiguousarray(past[::-1])
                    check(lib().dd_fir_reset_hist_f64(h, h64.ctypes.data_as(C.POINTER(C.c_double)), None),
                          "dd_fir_reset_hist_f64")
                self.__seeded = True
            return True
        check(lib().dd_fir_reset(h, _hip.DD_HIST_ZEROS, None, None), "dd_fir_reset")   # plain lfilter (filters.py:75)
        return False

    # -- public -------------------------------------------------------------------
    def applyOn(self, x):
        '''Apply the filter to a given array of signal

        Args:
            x: numpy array or device array

        Returns:
            filtered array of the same kind (complex64 / float64; the reference's
            SciPy path returns complex128 / float64 -- declared deviation Q6)
        '''
        from .comm import flush_all
        flush_all()
        host = not isinstance(x, DevArray)
        if host:
            a = np.asarray(x)
            # complex data: complex64 for the FIR kernels, complex128 kept for the float64 IIR recurrence
            d = DevArray.from_host(a, dtype=(_C64 if self.__isFIR else np.complex128) if np.iscomplexobj(a) else _F64)
        else:
            d = x
            if d.dtype == _hip.IQ8 and not self._fusable():
                from .comm import _convert
                d = _convert(d, _C64)
        if not self.__isFIR:
            out = self._apply_iir(d)
            return out.to_host() if host else out
        if self.__zeroPhase:
            out = _ops.filtfilt(self.__taps, d)
        elif d.dtype == _C64 or d.dtype == _hip.IQ8:
            out = _ops.fused(d, self, None, (1, 0), None)
        else:
            if d.dtype == _F32:
                from .comm import _convert
                d = _convert(d, _F64)
            carry = self._prepare_call()
            out = DevArray(d.n, _F64)
            check(lib().dd_fir_f64(self._handle(), d.ptr, out.ptr, d.n, 1 if carry else 0, None), "dd_fir_f64")
        return out.to_host() if host else out

    # -- IIR (butter): transposed direct form II recurrence on the device, float64 -------
    def _iir_handle(self):
        if self.__iir is None:
            _hip.require_gpu()
            b = np.atleast_1d(np.asarray(self.__b, dtype=np.float64))
            a = np.atleast_1d(np.asarray(self.__a, dtype=np.float64))
            n = max(len(a), len(b))
            b = np.r_[b, np.zeros(n - len(b))] / a[0]
            a = np.r_[a, np.zeros(n - len(a))] / a[0]
            # scipy.signal.lfilter_zi(b, a): steady-state state of the step response, used
            # unscaled by the reference (filters.py:45)
            comp = np.zeros((n - 1, n - 1))
            comp[0, :] = -a[1:]
            if n > 2:
                comp[1:, :-1] = np.eye(n - 2)
            zi = np.linalg.solve(np.eye(n - 1) - comp.T, b[1:] - a[1:] * b[0]) if n > 1 else np.zeros(0)
            self.__iir_zi = np.ascontiguousarray(zi)
            p = C.c_void_p()
            dp = C.POINTER(C.c_double)
            bb, aa = np.ascontiguousarray(b), np.ascontiguousarray(a)
            zarg = self.__iir_zi.ctypes.data_as(dp)
            check(lib().dd_iir_create(C.byref(p), bb.ctypes.data_as(dp), aa.ctypes.data_as(dp), n, zarg), "dd_iir_create")
            self.__iir = p
        return self.__iir

    def _apply_iir(self, d):
        if d.dtype == _C64 and not self.__zeroPhase and self.__initOut is None:
            # the IQ stream as the source hands it over (decode_funcube.py:160): complex64 in, complex128 out, no widened copy in between
            h = self._iir_handle()
            out = DevArray(d.n, np.complex128)
            check(lib().dd_iir_c64(h, d.ptr, out.ptr, d.n, 1 if self.__storeState else 0, None), "dd_iir_c64")
            return out
        if d.dtype == _C64:
            from .comm import _convert
            d = _convert(d, np.complex128)
        elif d.dtype == _F32:
            from .comm import _convert
            d = _convert(d, _F64)
        cplx = d.dtype == np.dtype(np.complex128)
        if not cplx and d.dtype != _F64:
            raise TypeError("unsupported dtype %s" % d.dtype)
        if self.__initOut is not None and not self.__zeroPhase:
            raise NotImplementedError("initOut with an IIR filter (lfiltic) is not supported on the device")
        h = self._iir_handle()
        out = DevArray(d.n, d.dtype)
        if self.__zeroPhase:
            check(lib().dd_iir_filtfilt_f64(h, d.ptr, out.ptr, d.n, 1 if cplx else 0, None), "dd_iir_filtfilt_f64")
        else:
            check(lib().dd_iir_f64(h, d.ptr, out.ptr, d.n, 1 if cplx else 0, 1 if self.__storeState else 0, None), "dd_iir_f64")
        return out

    @property
    def getA(self):
        ''':obj:`list`: Get 'a' of the filter'''
        return self.__a

    @property
    def getB(self):
        ''':obj:`list`: Get 'b' of the filter'''
        return self.__b


# ------------------------------------------------------------------ window designs
# Closed forms of scipy.signal.windows.{hamming,blackmanharris,gaussian}(n) (sym=True);
# the reference passes the raw window as the taps (filters.py:139,199,226).
def _cosine_sum(n, coeffs):
    if n == 1:
        return np.ones(1)
    ang = 2.0 * np.pi * np.arange(n) / (n - 1)
    w = np.zeros(n)
    for k, c in enumerate(coeffs):
        w += ((-1) ** k) * c * np.cos(k * ang)
    return w


class rollingAverage(filter):
    '''A simple rolling average filter (filters.py:95-114)'''

    def __init__(self, n=3, storeState=True, zeroPhase=False, initOut=None):
        self.__n = n
        super(rollingAverage, self).__init__([1.0 / n] * n, [1], storeState, zeroPhase, initOut)


class blackmanHarris(filter):
    '''Blackman Harris filter (filters.py:120-139)'''

    def __init__(self, n, storeState=True, zeroPhase=False, initOut=None):
        self.__n = n
        super(blackmanHarris, self).__init__(_cosine_sum(n, [0.35875, 0.48829, 0.14128, 0.01168]), [1],
                                             storeState, zeroPhase, initOut)


class hamming(filter):
    '''Hamming filter (filters.py:180-199)'''

    def __init__(self, n, storeState=True, zeroPhase=False, initOut=None):
        self.__n = n
        super(hamming, self).__init__(_cosine_sum(n, [0.54, 0.46]), [1], storeState, zeroPhase, initOut)


class gaussian(filter):
    '''Gaussian filter (filters.py:205-226)'''

    def __init__(self, n, sigma, storeState=True, zeroPhase=False, initOut=None):
        self.__n = n
        self.__sigma = sigma
        k = np.arange(n) - (n - 1.0) / 2.0
        super(gaussian, self).__init__(np.exp(-k ** 2 / (2.0 * sigma * sigma)), [1], storeState, zeroPhase, initOut)


class butter(filter):
    '''Butterworth filter (filters.py:232-273).  IIR: a float64 transposed direct form II
    recurrence on the device -- one lane per component for short (audio-rate) inputs, the
    block-parallel form (dd_fir.hip: block end states, two-level scan of the start states,
    re-run) from 4096 samples up, which is what full-rate IQ through a butter takes
    (decode_funcube.py:160,230).'''

    def __init__(self, Fs, cutoffA, cutoffB=None, n=6, typeFlt=constants.FLT_LP, storeState=True,
                 zeroPhase=False, initOut=None):
        import scipy.signal as signal     # coefficient design only
        if (typeFlt == constants.FLT_BP or typeFlt == constants.FLT_BS) and cutoffB is None:
            raise ValueError("CutoffB must be given")
        nyq = 0.5 * Fs
        if typeFlt == constants.FLT_LP:
            b, a = signal.butter(n, cutoffA / nyq, btype='lowpass')
        elif typeFlt == constants.FLT_HP:
            b, a = signal.butter(n, cutoffA / nyq, btype='highpass')
        elif typeFlt == constants.FLT_BP:
            b, a = signal.butter(n, [cutoffA / nyq, cutoffB / nyq], btype='bandpass')
        elif typeFlt == constants.FLT_BS:
            b, a = signal.butter(n, [cutoffA / nyq, cutoffB / nyq], btype='bandstop')
        else:
            raise ValueError("Invalid filter type")
        super(butter, self).__init__(b, a, storeState, zeroPhase, initOut)


class remez(filter):
    '''Remez (Parks-McClellan) band filter (filters.py:279-314)'''

    def __init__(self, Fs, bands, gains, ntaps=128, storeState=True, zeroPhase=False, initOut=None):
        import scipy.signal as signal     # coefficient design only
        if len(bands) == 0:
            raise ValueError("Atleast one band must be given")
        if bands[-1][1] >= (Fs / 2):
            raise ValueError("Last band must end before (Fs/2)Hz")
        flat = []
        for i in bands:
            flat.extend(i)
        if not len(flat) == 2 * len(gains):
            raise ValueError("Invalid bands/gains values")
        super(remez, self).__init__(signal.remez(ntaps, flat, gains, fs=Fs), [1], storeState, zeroPhase, initOut)
