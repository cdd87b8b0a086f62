"""
FM demodulation -- drop-in for the reference's directdemod/demod_fm.py:12-51.

``demod(sig)`` = angle(sig[1:] * conj(sig[:-1])) with a one-sample carry between
calls (first call returns L-1 values, later calls L; quirk Q3).  On the device the
carried sample is the last FIR output of the previous chunk, kept in HBM inside the
C handle; inside a commSignal chain the discriminator is the epilogue of the fused
FIR kernel (no intermediate array).
"""
import ctypes as C

import numpy as np

from . import _hip
from ._hip import DevArray, check, lib


_FM_POOL = []


class demod_fm():
    '''
    Object for FM demodulation
    '''

    def __init__(self, storeState=True):
        '''Args:
            storeState (:obj:`bool`): Store state? Helps if signal is chunked
        '''
        self.__storeState = storeState
        self.__h = None
        self.__rec_has_last = False      # state as seen at call (record) time
        self.__dev_has_last = False      # state as seen at execution time

    # -- handle management ------------------------------------------------------
    def _handle(self):
        if self.__h is None:
            _hip.require_gpu()
            if _FM_POOL:                     # a handle of a collected demodulator, back in its first-call state (see filters._FIR_POOL)
                p = _FM_POOL.pop()
                check(lib().dd_fm_reset(p), "dd_fm_reset")
            else:
                p = C.c_void_p()
                check(lib().dd_fm_create(C.byref(p)), "dd_fm_create")
            self.__h = p
        return self.__h

    def __del__(self):
        try:
            if self.__h is not None:
                if len(_FM_POOL) < 16:
                    _FM_POOL.append(self.__h)
                else:
                    lib().dd_fm_destroy(self.__h)
                self.__h = None
        except Exception:
            pass

    # -- bookkeeping used by commSignal's deferred execution --------------------
    def _note_call(self, n):
        """length of the output for an input of n samples, at call time"""
        if n < 1 and self.__storeState:
            raise IndexError("index -1 is out of bounds for axis 0 with size 0")   # demod_fm.py:44/48
        if self.__storeState and self.__rec_has_last:
            return n
        if self.__storeState:
            self.__rec_has_last = True
        return max(0, n - 1)

    def _prepare_call(self):
        if not self.__storeState:
            check(lib().dd_fm_reset(self._handle()), "dd_fm_reset")
            self.__dev_has_last = False

    def _carries(self):
        return bool(self.__storeState)

    def _dev_has_last(self):
        return self.__dev_has_last

    def _after_call(self):
        self.__dev_has_last = bool(self.__storeState)

    # -- public -------------------------------------------------------------------
    def _demod_device(self, x):
        if x.dtype != np.dtype(np.complex64):
            raise TypeError("demod_fm expects complex IQ samples")
        h = self._handle()
        self._prepare_call()
        n_expect = max(0, x.n - (0 if self.__dev_has_last else 1))
        out = DevArray(n_expect, np.float32)
        no = C.c_int64(0)
        check(lib().dd_fm_discrim_c64(h, x.ptr, out.ptr, x.n, 1 if self.__storeState else 0,
                                      C.byref(no), None), "dd_fm_discrim_c64")
        assert no.value == n_expect
        self._after_call()
        return out

    def demod(self, sig):
        '''FM demod a given complex IQ array

        Args:
            sig: numpy array (complex) or device array

        Returns:
            numpy array float64 for numpy input (like the reference), device float32
            array for device input
        '''
        if isinstance(sig, DevArray):
            from .comm import flush_all
            flush_all()
            self._note_call(sig.n)
            return self._demod_device(sig)
        from .comm import flush_all
        flush_all()
        a = np.asarray(sig)
        self._note_call(len(a))
        out = self._demod_device(DevArray.from_host(a, dtype=np.complex64))
        return out.to_host().astype(np.float64)
