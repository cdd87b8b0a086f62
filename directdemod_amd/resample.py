"""
Polyphase rational resampler (BASELINE north_star; config 3: "... + FM demod + polyphase resample to
11.025 kS/s").  The reference has no counterpart -- ``commSignal.bwLim(t, strict=True)`` is an FFT resample of the
whole chunk (comm.py:110-116), which this package reproduces exactly under the same name -- so this is a
build-defined EXTENSION.  It follows SciPy's ``scipy.signal.resample_poly`` (same low-pass design, same output
alignment and length) and, unlike both the reference's strict ``bwLim`` and ``resample_poly`` itself, carries its
state from chunk to chunk: the concatenated outputs of a chunk loop equal ``resample_poly`` of the whole stream
(no border effects per chunk; SURVEY.md App. B: the reference's per-chunk resample yields 11 020 instead of 11 025
samples for 1 s in 10 chunks).

    rs = resample.polyResampler(200000, 11025)            # once, outside the chunk loop (like a filter)
    for ...: audio.extend(sig.filter(...).bwLim(200000).funcApply(fm.demod).resamplePoly(rs))
    audio.extend(rs.flush())                          # the tail (the filter's look-ahead)
"""
import ctypes as C
import math

import numpy as np

from . import _hip
from ._hip import DevArray, check, lib


def design(up, down, window=("kaiser", 5.0)):
    """(up, down in lowest terms, front-padded taps, n_pre_remove) exactly as scipy.signal.resample_poly builds them"""
    from scipy.signal import firwin
    g = math.gcd(int(up), int(down))
    up, down = int(up) // g, int(down) // g
    half_len = 10 * max(up, down)
    h = firwin(2 * half_len + 1, 1.0 / max(up, down), window=window) * up
    n_pre_pad = down - half_len % down
    return up, down, np.concatenate((np.zeros(n_pre_pad), h)), (half_len + n_pre_pad) // down


class polyResampler:
    '''Rational-rate resampler sampRate -> tsampRate with carried state (create once, outside the chunk loop)'''

    def __init__(self, sampRate, tsampRate, window=("kaiser", 5.0)):
        if int(sampRate) <= 0 or int(tsampRate) <= 0:
            raise ValueError("sampling rates must be positive")
        self.inRate, self.outRate = int(sampRate), int(tsampRate)
        self.up, self.down, self.taps, self.n_pre_remove = design(self.outRate, self.inRate, window)
        self.__h = None

    def _handle(self):
        if self.__h is None:
            _hip.require_gpu()
            t = np.ascontiguousarray(self.taps, dtype=np.float64)
            p = C.c_void_p()
            check(lib().dd_rpoly_create(C.byref(p), t.ctypes.data_as(C.POINTER(C.c_double)), len(t), self.up, self.down,
                                        self.n_pre_remove), "dd_rpoly_create")
            self.__h = p
        return self.__h

    def __del__(self):
        try:
            if self.__h is not None:
                lib().dd_rpoly_destroy(self.__h)
        except Exception:
            pass

    def _run(self, d, n, flush):
        h = self._handle()
        cnt = int(lib().dd_rpoly_out_count(h, n, 1 if flush else 0))
        out = DevArray(cnt, np.float64)
        got = C.c_int64(0)
        check(lib().dd_rpoly_process(h, d.ptr if d is not None else None, n, 1 if flush else 0, out.ptr, C.byref(got), None),
              "dd_rpoly_process")
        return out

    def applyOn(self, x):
        '''the output samples that this chunk completes (numpy in -> numpy out, device array in -> device array out)'''
        from .comm import flush_all, _convert
        flush_all()
        host = not isinstance(x, DevArray)
        if host:
            a = np.asarray(x)
            if np.iscomplexobj(a):
                raise NotImplementedError("polyResampler is implemented for real signals (the audio-rate stages)")
            d = DevArray.from_host(a, dtype=np.float64)
        else:
            d = x if x.dtype == np.dtype(np.float64) else _convert(x, np.dtype(np.float64))
        out = self._run(d, d.n, False)
        return out.to_host() if host else out

    def flush(self, as_host=True):
        '''the remaining outputs: ceil(n_in * up / down) in all have then been produced'''
        out = self._run(None, 0, True)
        return out.to_host() if as_host else out

    def reset(self):
        if self.__h is not None:
            check(lib().dd_rpoly_reset(self.__h), "dd_rpoly_reset")
