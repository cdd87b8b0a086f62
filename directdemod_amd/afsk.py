"""
AFSK1200 correlator front end on the device -- the part of the reference's
decode_afsk1200.getMsg (decode_afsk1200.py:99-158) that is a pure-Python double
loop there: quadrature correlation against the mark (1200 Hz) and space (2200 Hz)
tones over one baud, the power difference ("binary filter"), and the bit-edge
detector.  The frame logic behind it (peak pick, NRZI, bit de-stuffing, CRC) is
host code in the reference and is not rebuilt here.

The FM audio that feeds it comes from the same fused chain as every other decoder
(offsetFreq -> blackmanHarris(151) -> bwLim(bw) -> demod_fm -> butter band-pass,
decode_afsk1200.py:67-98).
"""
import ctypes as C

import numpy as np

from . import _hip
from ._hip import DevArray, check, lib

_F64 = np.dtype(np.float64)

BAUDRATE = 1200             # decode_afsk1200.py:31-33
MARK_FREQUENCY = 1200
SPACE_FREQUENCY = 2200


def correlator_tables(bw, baud=BAUDRATE, mark=MARK_FREQUENCY, space=SPACE_FREQUENCY):
    """decode_afsk1200.py:99-123 -> (tables[4, buffer_size] = mark_i, mark_q, space_i, space_q; samples per baud)"""
    buffer_size = int(np.round(bw / baud))
    samples_per_baud = bw // baud
    i = np.arange(buffer_size)
    mark_angle = (i * 1.0 / bw) / (1 / mark) * 2 * np.pi
    space_angle = (i * 1.0 / bw) / (1 / space) * 2 * np.pi
    return np.ascontiguousarray([np.cos(mark_angle), np.sin(mark_angle), np.cos(space_angle), np.sin(space_angle)],
                                dtype=np.float64), int(samples_per_baud)


def _dev(x):
    if isinstance(x, DevArray):
        if x.dtype != _F64:
            raise TypeError("float64 device array expected, got %s" % x.dtype)
        return x, False
    return DevArray.from_host(np.asarray(x, dtype=np.float64).ravel()), True


def binary_filter(sig, bw=22050, baud=BAUDRATE, mark=MARK_FREQUENCY, space=SPACE_FREQUENCY):
    """mark-minus-space correlator power per sample (decode_afsk1200.py:126-141).
    NumPy in -> NumPy out; DevArray in -> DevArray out."""
    _hip.require_gpu()
    tables, _ = correlator_tables(bw, baud, mark, space)
    d, host = _dev(sig)
    out = DevArray(d.n, _F64)
    check(lib().dd_afsk_binary_filter_f64(d.ptr, d.n, tables.ctypes.data_as(C.POINTER(C.c_double)), tables.shape[1],
                                          out.ptr, None), "dd_afsk_binary_filter_f64")
    return out.to_host() if host else out


def bit_edges(bf, samples_per_baud):
    """np.correlate(np.sign(bf), edge kernel, 'same') / samples_per_baud (decode_afsk1200.py:147-156)"""
    _hip.require_gpu()
    d, host = _dev(bf)
    out = DevArray(d.n, _F64)
    check(lib().dd_afsk_edges_f64(d.ptr, d.n, int(samples_per_baud), out.ptr, None), "dd_afsk_edges_f64")
    return out.to_host() if host else out
