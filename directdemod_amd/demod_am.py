"""
AM demodulation -- drop-in for the reference's directdemod/demod_am.py:12-29:
``demod(sig) = abs(hilbert(sig))`` over the array it is given (the NOAA decoder
feeds it fixed 240 000-sample blocks, decode_noaa.py:647-653).  Audio-rate stage,
computed in float64 on the device (hipFFT transforms + hand-written spectrum mask /
magnitude kernels) so that sync index picks stay bit-exact.
"""
import numpy as np

from . import _ops
from ._hip import DevArray


class demod_am():
    '''
    AM demodulation by hilbert's transform
    '''

    def demod(self, sig):
        '''Args:
            sig: real numpy array or device array

        Returns:
            envelope; numpy float64 for numpy input, device float64 for device input
        '''
        from .comm import flush_all
        flush_all()
        if isinstance(sig, DevArray):
            return _ops.am_envelope(sig)
        a = np.asarray(sig)
        if np.iscomplexobj(a):
            raise TypeError("demod_am expects a real signal")   # scipy.signal.hilbert: "x must be real."
        return _ops.am_envelope(DevArray.from_host(a, dtype=np.float64)).to_host()

    def demod_blocks(self, sig, block=60000 * 4):
        '''Envelope in independent fixed blocks laid out by the chunker rule, all
        blocks in one batched device call (decode_noaa.__getAM, decode_noaa.py:631-657).'''
        from .comm import flush_all
        flush_all()
        if isinstance(sig, DevArray):
            return _ops.am_envelope(sig, block)
        return _ops.am_envelope(DevArray.from_host(np.asarray(sig), dtype=np.float64), block).to_host()


class demod_amFLT():
    '''
    AM demodulation by low pass filter (demod_am.py:35-62): butter(Fs, cutoff) applied to |sig|,
    the filter state carried from call to call like the reference's.  Runs on the device
    (magnitude kernel + the float64 IIR recurrence, block-parallel for long inputs).
    '''

    def __init__(self, Fs, cutoff):
        from . import filters
        self.__filter = filters.butter(Fs, cutoff)

    def demod(self, sig):
        '''Args:
            sig: numpy array or device array (real or complex)

        Returns:
            demodulated signal, float64; numpy for numpy input, device array for device input
        '''
        import ctypes as C
        from . import _hip
        from .comm import flush_all
        flush_all()
        host = not isinstance(sig, DevArray)
        if host:
            a = np.asarray(sig)
            d = DevArray.from_host(a, dtype=np.complex128 if np.iscomplexobj(a) else np.float64)
        else:
            d = sig
        kind = {np.dtype(np.float64): 0, np.dtype(np.complex128): 1, np.dtype(np.complex64): 2}.get(d.dtype)
        if kind is None:
            if d.dtype == np.dtype(np.float32):
                from .comm import _convert
                d, kind = _convert(d, np.float64), 0
            else:
                raise TypeError("unsupported dtype %s" % d.dtype)
        mag = DevArray(d.n, np.float64)
        _hip.check(_hip.lib().dd_abs_f64(d.ptr, kind, mag.ptr, d.n, None), "dd_abs_f64")
        out = self.__filter.applyOn(mag)
        return out.to_host() if host else out
