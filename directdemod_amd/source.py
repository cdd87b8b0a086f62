"""
IQ sources -- feeders of the hot path (reference:
directdemod/source.py:53-324).  ``read(a, b)`` keeps the reference's contract
(complex64, I + jQ - (127.5 + 127.5j), ValueError on bad indices, ``limitData``);
``read_device(a, b)`` is the MI355X ingest: the raw interleaved uint8 pairs cross
PCIe (2 B/sample instead of 8) through a pinned staging buffer and are widened on
the device (dd_u8iq_to_c64), or consumed directly by the fused kernel
(DD_CHAIN_U8_INPUT).
"""
import ctypes as C

import numpy as np

from . import _hip, constants
from ._hip import DevArray, check, lib


class _u8source:
    """Common logic over an interleaved uint8 [N, 2] array (memmap or in memory)."""

    def __init__(self, data_u8_n2, sampFreq, sourceType):
        self._data = data_u8_n2
        self.__sampFreq = sampFreq
        self.__sourceType = sourceType
        self.__offset = 0
        self.__actualLength = data_u8_n2.shape[0]
        self.__length = data_u8_n2.shape[0]

    @property
    def sampFreq(self):
        ''':obj:`int`: get sampling freq of source'''
        return self.__sampFreq

    @property
    def sourceType(self):
        ''':obj:`int`: get source type'''
        return self.__sourceType

    @property
    def length(self):
        ''':obj:`int`: get source length'''
        return self.__length

    def _range(self, fromIndex, toIndex):
        if toIndex is None:
            toIndex = fromIndex + 1
        if fromIndex < 0 or toIndex < 0 or fromIndex >= self.length or toIndex > self.length:
            raise ValueError("fromIndex and toIndex have invalid values")          # source.py:114-115
        return fromIndex + self.__offset, toIndex + self.__offset

    def read(self, fromIndex, toIndex=None):
        '''Complex IQ samples in a numpy array (host), like the reference's read'''
        a, b = self._range(fromIndex, toIndex)
        d = np.asarray(self._data[a:b])
        out = np.empty(b - a, dtype=np.complex64)
        out.real = d[:, 0]
        out.imag = d[:, 1]
        out -= np.complex64(127.5 + 127.5j)
        return out

    def read_raw_u8(self, fromIndex, toIndex=None):
        '''the raw interleaved uint8 pairs (no conversion): feed for the fused u8 ingest'''
        a, b = self._range(fromIndex, toIndex)
        return np.ascontiguousarray(self._data[a:b]).reshape(-1)

    def raw_view(self, fromIndex, toIndex=None):
        '''the raw pairs as a C-contiguous uint8 array WITHOUT a copy where the backing store allows (array, memmap)'''
        a, b = self._range(fromIndex, toIndex)
        v = self._data[a:b]
        return v if v.flags["C_CONTIGUOUS"] else np.ascontiguousarray(v)

    def read_raw_u8_into(self, dst, fromIndex, toIndex=None):
        '''the raw pairs copied straight into a caller buffer (e.g. a pinned staging slot): one host copy'''
        a, b = self._range(fromIndex, toIndex)
        dst[:2 * (b - a)] = self._data[a:b].reshape(-1)

    def read_device(self, fromIndex, toIndex=None):
        '''Complex64 samples as a device array: 2 B/sample over PCIe, widened in HBM'''
        raw = self.read_raw_u8(fromIndex, toIndex)
        n = raw.size // 2
        d_raw = DevArray.from_host(raw, dtype=np.uint8)
        out = DevArray(n, np.complex64)
        check(lib().dd_u8iq_to_c64(d_raw.ptr, out.ptr, n, None), "dd_u8iq_to_c64")
        _hip.sync()
        return out

    # -- the recording resident in HBM as raw pairs (2 B/sample): uploaded once, page by page on first use --
    _PAGE = 1 << 22                     # samples per upload page (8 MiB)

    def read_device_raw(self, fromIndex, toIndex=None):
        '''The samples as a device array of raw uint8 pairs (dtype _hip.IQ8), a view into a buffer that holds
        the whole recording on the device.  Pages are uploaded the first time they are touched and stay: the
        fused kernels widen the pairs themselves (DD_CHAIN_U8_INPUT), and a later pass over the same recording
        (the accurate-sync windows after the crude sync) costs no host copy and no PCIe traffic.
        Returns None when the recording is larger than DD_RESIDENT_BYTES (default 64 GiB).'''
        import os
        a, b = self._range(fromIndex, toIndex)
        total = self._data.shape[0]
        if 2 * total > int(os.environ.get("DD_RESIDENT_BYTES", str(64 << 30))):
            return None
        if getattr(self, "_res", None) is None:
            _hip.wait_copy_warmup()
            self._res = DevArray(total, _hip.IQ8)
            self._res_pages = np.zeros((total + self._PAGE - 1) // self._PAGE, dtype=bool)
        for pg in range(a // self._PAGE, (b - 1) // self._PAGE + 1):
            if not self._res_pages[pg]:
                lo, hi = pg * self._PAGE, min(total, (pg + 1) * self._PAGE)
                src = self._data[lo:hi]
                if not src.flags["C_CONTIGUOUS"]:
                    src = np.ascontiguousarray(src)
                check(lib().dd_memcpy_h2d(self._res.ptr + 2 * lo, src.ctypes.data, 2 * (hi - lo), None), "h2d")
                self._res_pages[pg] = True
        _hip.sync()
        return self._res.view(a, b - a)

    def resident_raw(self, fromIndex, toIndex):
        '''(device pointer of the resident recording, absolute index of fromIndex) with [fromIndex, toIndex) uploaded,
        or None'''
        v = self.read_device_raw(fromIndex, toIndex)
        if v is None:
            return None
        a, _ = self._range(fromIndex, toIndex)
        return self._res, a

    def limitData(self, initOffset=None, finalLimit=None):
        '''Limit source data (source.py:120-138)'''
        self.__offset = initOffset if initOffset is not None else 0
        if finalLimit is not None:
            self.__length = finalLimit - self.__offset
        else:
            self.__length = self.__actualLength


class IQarray(_u8source):
    '''In-memory uint8 [N, 2] recording (synthetic inputs, tests, benchmarks)'''

    def __init__(self, raw_u8_n2, sampFreq):
        raw = np.asarray(raw_u8_n2, dtype=np.uint8)
        if raw.ndim != 2 or raw.shape[1] != 2:
            raise TypeError("expected a uint8 array of shape [N, 2]")
        super().__init__(raw, sampFreq, constants.SOURCE_IQWAV)


def _wav_data_chunk(filename):
    """Walk the RIFF chunks of a WAV file to its ``fmt `` and ``data`` chunks, like
    scipy.io.wavfile.read does for the reference (source.py:68): SDRSharp recordings carry an
    ``auxi`` chunk (and others may carry ``LIST``) in front of ``data``, so the samples do not
    start at byte 44.  Returns (sample rate, channels, bits, data offset, data bytes)."""
    import os
    import struct
    size = os.path.getsize(filename)
    with open(filename, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError("%s is not a little-endian RIFF/WAVE file" % filename)
        fmt = None
        pos = 12
        while pos + 8 <= size:
            f.seek(pos)
            cid, csz = struct.unpack("<4sI", f.read(8))
            body = pos + 8
            if cid == b"fmt ":
                raw = f.read(min(csz, 16))
                if len(raw) < 16:
                    raise ValueError("%s: truncated fmt chunk" % filename)
                tag, ch, rate, _, _, bits = struct.unpack("<HHIIHH", raw)
                fmt = (rate, ch, bits, tag)
            elif cid == b"data":
                if fmt is None:
                    raise ValueError("%s: data chunk before fmt chunk" % filename)
                return fmt[0], fmt[1], fmt[2], body, min(csz, size - body)
            pos = body + csz + (csz & 1)          # chunks are word aligned
    raise ValueError("%s: no data chunk" % filename)


class IQwav(_u8source):
    '''8-bit stereo IQ.wav (SDRSharp style).  The ``data`` chunk is located by walking the RIFF
    chunks and the sample rate comes from ``fmt `` unless given -- what scipy.io.wavfile.read
    gives the reference (source.py:66-72).'''

    def __init__(self, filename, givenSampFreq=None):
        rate, ch, bits, off, nbytes = _wav_data_chunk(filename)
        if ch != 2 or bits != 8:
            raise TypeError("IQ.wav must be 8-bit stereo (I, Q); got %d channel(s) of %d bits" % (ch, bits))
        mm = np.memmap(filename, dtype=np.uint8, mode="r", offset=off, shape=(nbytes,))
        n = nbytes // 2
        super().__init__(mm[:2 * n].reshape(n, 2), givenSampFreq if givenSampFreq is not None else rate,
                         constants.SOURCE_IQWAV)


class IQdat(_u8source):
    '''Raw interleaved uint8 I,Q file (rtl_sdr style)'''

    def __init__(self, filename, givenSampFreq=None):
        mm = np.memmap(filename, dtype=np.uint8, mode="r")
        n = mm.shape[0] // 2
        super().__init__(mm[:2 * n].reshape(n, 2),
                         givenSampFreq if givenSampFreq is not None else constants.IQ_SDRSAMPRATE,
                         constants.SOURCE_IQDAT)


class IQwavAlt(_u8source):
    '''The reference's alternative IQ.wav reader (source.py:237-324): memmap behind a 44-byte
    header, the sample rate NOT taken from the header (IQ_SDRSAMPRATE unless given).'''

    def __init__(self, filename, givenSampFreq=None):
        mm = np.memmap(filename, dtype=np.uint8, mode="r", offset=44)
        n = mm.shape[0] // 2
        super().__init__(mm[:2 * n].reshape(n, 2),
                         givenSampFreq if givenSampFreq is not None else constants.IQ_SDRSAMPRATE,
                         constants.SOURCE_IQWAV)
