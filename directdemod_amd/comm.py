"""
commSignal -- device-resident signal container, drop-in for the reference's
directdemod/comm.py:15-181 (same constructor, fluent methods and properties).

MI355X design.  The reference runs every fluent call as a separate full-array
NumPy/SciPy pass.  Here the samples live in HBM and the calls that make up the
per-sample hot path --

    .offsetFreq(f)  .filter(fir)  .bwLim(rate)  .funcApply(fm.demod)
    (decode_noaa.py:623, decode_fm.py:64-68, tutorial/3_chunking.py:30-37)

-- are *recorded* and executed as ONE fused HIP kernel (dd_fused_process: NCO ->
LDS-staged FIR -> only the kept outputs -> atan2 discriminator) the moment a result
is needed.  All bookkeeping the reference does at call time (chunker variables,
lengths, sample rates, first-chunk quirks) is still done at call time, so the
observable behaviour is the reference's; only the arithmetic is deferred.  Pending
work is kept in one global FIFO so operator state (filter history, last FM sample)
always advances in call order.
"""
import numpy as np

from . import _hip, constants
from ._hip import DevArray

_C64 = np.dtype(np.complex64)
_F32 = np.dtype(np.float32)
_F64 = np.dtype(np.float64)
_IQ8 = _hip.IQ8

# FIFO of commSignal objects that still have recorded, un-executed operations
_pending = []


def flush_all():
    """Execute every recorded operation, oldest signal first.  A run of pending signals that are the chunks of ONE chunk
    loop over a device-resident recording -- consecutive views of one buffer through the same filter / demodulator with
    the chunker variables following on -- executes as one chunk-list call (dd_fused_process_chunks: one launch,
    bit-identical to chunk by chunk; the strict bwLim that may end each chunk as one dd_resample_fft_chunks)."""
    while _pending:
        k = _batch_len()
        if k >= 2:
            batch = _pending[:k]
            del _pending[:k]
            _run_batch(batch)
        else:
            s = _pending.pop(0)
            s._run_ops()


def _chunk_pattern(s):
    """(nco op or None, filter, (M, off), fm or None, resample num or None) when the recorded operations of `s` are exactly
    [offsetFreq] filter bwLim [demod_fm] [bwLim strict] on device-resident IQ samples; None otherwise"""
    ops = s._ops
    if s._dev is None or s._dev.dtype not in (_C64, _IQ8) or s._dev.n == 0:
        return None
    i = 0
    nco = None
    if i < len(ops) and ops[i][0] == "nco":
        nco = ops[i]
        i += 1
    if not (i < len(ops) and ops[i][0] == "fir"):
        return None
    filt = ops[i][1]
    i += 1
    if not (i < len(ops) and ops[i][0] == "decim" and ops[i][1] > 1):
        return None
    decim = (ops[i][1], ops[i][2])
    i += 1
    fm = None
    if i < len(ops) and ops[i][0] == "fm":
        fm = ops[i][1]
        i += 1
    rs = None
    if i < len(ops) and ops[i][0] == "resample":
        rs = ops[i][1]
        i += 1
    if i != len(ops) or not filt._carries() or (fm is not None and not fm._carries()):
        return None
    return nco, filt, decim, fm, rs


def _batch_len():
    first = _chunk_pattern(_pending[0])
    if first is None:
        return 1
    nco0, filt, (M, off), fm, rs0 = first
    k = 1
    prev = _pending[0]
    nco_prev, off_prev = nco0, off
    while k < len(_pending):
        cur = _pending[k]
        pat = _chunk_pattern(cur)
        if pat is None:
            break
        nco, f2, (M2, off2), fm2, rs = pat
        n_prev = prev._dev.n
        if f2 is not filt or fm2 is not fm or M2 != M or (rs is None) != (rs0 is None) or (nco is None) != (nco0 is None):
            break
        if cur._dev.dtype != prev._dev.dtype or cur._dev.ptr != prev._dev.ptr + n_prev * prev._dev.dtype.itemsize:
            break                                              # not the next samples of the same buffer
        if nco is not None and (nco[1] != nco_prev[1] or nco[2] != nco_prev[2] + n_prev):
            break                                              # "freqoffset" does not follow on (comm.py:75-76)
        if off2 != (M - (n_prev - off_prev) % M) % M:
            break                                              # "bwlim" phase does not follow on (comm.py:123-125)
        prev, nco_prev, off_prev = cur, nco, off2
        k += 1
    return k


def _run_batch(sigs):
    import ctypes as C
    from . import _ops
    nco0, filt, (M, off0), fm, rs0 = _chunk_pattern(sigs[0])
    k = len(sigs)
    fir_h = filt._handle()
    filt._prepare_call()
    fm_h = None
    has_last = True
    if fm is not None:
        fm_h = fm._handle()
        fm._prepare_call()
        has_last = fm._dev_has_last()
    lens = [s._dev.n for s in sigs]
    bounds = [0]
    for n in lens:
        bounds.append(bounds[-1] + n)
    # expected outputs per chunk (the same arithmetic _ops.fused asserts per call)
    expect, off = [], off0
    for i, n in enumerate(lens):
        kept = len(range(off, n, M))
        e = kept
        if fm is not None:
            e = max(0, kept - (0 if has_last else 1))
            has_last = True
        expect.append(e)
        off = (M - (n - off) % M) % M
    x0 = sigs[0]._dev
    out = DevArray(max(1, sum(expect)), _F32 if fm is not None else _C64)
    flags = (_hip.DD_CHAIN_FORCE_DIRECT if _ops.FORCE_DIRECT else 0) | (_hip.DD_CHAIN_U8_INPUT if x0.dtype == _IQ8 else 0) | \
            (_hip.DD_CHAIN_TIGHT if getattr(filt, "tight", False) else 0)
    nout = (C.c_int64 * k)()
    _hip.check(_hip.lib().dd_fused_process_chunks(fir_h, fm_h, x0.ptr, out.ptr, (C.c_int64 * (k + 1))(*bounds), k,
                                                  1 if nco0 is not None else 0, nco0[1] if nco0 is not None else 0,
                                                  nco0[2] if nco0 is not None else 0, M, off0, flags, nout, None),
               "dd_fused_process_chunks")
    assert list(nout) == expect, (list(nout), expect)
    if fm is not None:
        fm._after_call()
    offs = [0]
    for e in expect:
        offs.append(offs[-1] + e)
    if rs0 is not None:
        nums = [_chunk_pattern(s)[4] for s in sigs]
        res, ooff = _ops.resample_fft_chunks(out, offs[:-1], expect, nums)
        pieces = [res.view(ooff[i], nums[i]) for i in range(k)]
    else:
        pieces = [out.view(offs[i], expect[i]) for i in range(k)]
    for s, piece in zip(sigs, pieces):
        s._ops = []
        s._dev = piece
        s._host = None
        s._cap = None


class commSignal:
    '''
    Stores a signal (in device memory) and its sampling rate
    '''

    def __init__(self, sampRate, sig=np.array([]), chunker=None):
        '''Initialize the object

        Args:
            sampRate (:obj:`int`): sampling rate in Hz, forced to int (comm.py:34)
            sig (:obj:`numpy array` or device array, optional): one dimensional
            chunker (:obj:`chunker`, optional): chunking object, if processed in chunks
        '''
        self.__chunker = chunker
        self.__len = len(sig)
        self.__sampRate = int(sampRate)
        if self.__sampRate <= 0:
            raise ValueError("The sampling rate must be greater than zero")
        self._ops = []
        self._host = None
        self._dev = None
        self._cap = None            # growable device buffer used by extend()
        self._lazy = []             # signals extend()ed while their own operations were still pending (appended at materialise)
        self._into = None           # the container this signal has been lazily extend()ed into
        self._store(sig, copy=True)

    # ------------------------------------------------------------------ storage
    def _store(self, sig, copy):
        if isinstance(sig, DevArray):
            self._dev = sig
            self._host = None
        else:
            a = np.array(sig) if copy else np.asarray(sig)      # ctor copies (comm.py:38)
            if a.ndim == 0 or not a.size == a.shape[0]:
                raise TypeError("The signal array must be 1-D")
            self._host = a
            self._dev = None
        self._cap = None
        self._ops = []
        self._lazy = []

    def _device(self, want=None):
        """Device copy of the stored array.  Complex -> complex64, real -> float64
        unless the array already is float32 on the device (FM output)."""
        if self._dev is None:
            a = self._host
            if np.iscomplexobj(a):
                dt = _C64
            else:
                dt = _F64
            self._dev = DevArray.from_host(a, dtype=dt)
        if want is not None and self._dev.dtype != np.dtype(want):
            self._dev = _convert(self._dev, want)
            self._host = None
        return self._dev

    # ------------------------------------------------------------------ properties
    @property
    def length(self):
        ''':obj:`int`: get length of signal'''
        return self.__len

    @property
    def sampRate(self):
        ''':obj:`int`: get sampling rate of signal'''
        return self.__sampRate

    @property
    def signal(self):
        ''':obj:`numpy array`: the samples (downloads from the device when needed).
        Device float32 results (FM output) are handed out as float64 like the
        reference's; complex stays complex64 (declared deviation Q6: the reference
        holds complex128 after ``filter``).'''
        self._materialise()
        if self._host is None:
            if self._dev.dtype == _IQ8:
                self._dev = _convert(self._dev, _C64)
            a = self._dev.to_host()
            if a.dtype == _F32:
                a = a.astype(np.float64)
            self._host = a
        return self._host

    @property
    def device_signal(self):
        """The samples as a device array (no download)."""
        self._materialise()
        return self._device()

    def _materialise(self):
        if self._ops or self._lazy:
            flush_all()
            lazies, self._lazy = self._lazy, []
            if lazies and self._adopt(lazies):
                return
            for sig in lazies:
                sig._into = None
                self._append(sig)

    def _adopt(self, lazies):
        """An empty container whose pending chunks turned out as consecutive pieces of ONE device buffer (the chunk-list call
        writes a chunk loop's outputs back to back) takes a view of that stretch instead of copying piece by piece.  No
        operation of this package writes into its input, and growing the container later copies (`_append`), so the shared
        samples never change under either owner.  (Seven to sixteen device-to-device copies per chunk loop: 5 us each on the
        device and as much again on the host -- profiles/r04_noaa_timeline.txt.)"""
        if self._phys_len() != 0 or self._cap is not None:
            return False
        first = lazies[0]._dev
        if first is None or first._base is None or first.dtype == _IQ8:
            return False
        end = first.ptr + first.nbytes
        for s in lazies[1:]:
            d = s._dev
            if d is None or d._base is not first._base or d.dtype != first.dtype or d.ptr != end:
                return False
            end += d.nbytes
        base = first._base
        self._dev = base.view((first.ptr - base.ptr) // first.dtype.itemsize, (end - first.ptr) // first.dtype.itemsize)
        self._host = None
        for s in lazies:
            s._into = None
        return True

    def _settle(self):
        """before this signal changes: if it sits in a container's lazy list, let the container take its samples first"""
        if self._into is not None:
            self._into._materialise()

    # ------------------------------------------------------------------ hot path (recorded)
    def _record(self, op):
        self._settle()
        if self._lazy:
            self._materialise()
        if not self._ops:
            _pending.append(self)
        self._ops.append(op)

    def offsetFreq(self, freqOffset):
        '''Offset signal by a frequency by multiplying a complex envelope (comm.py:63-78)'''
        offset = 0
        if self.__chunker is not None:
            offset = self.__chunker.get(constants.CHUNK_FREQOFFSET, 0)
            self.__chunker.set(constants.CHUNK_FREQOFFSET, offset + self.length)
        if np.ndim(freqOffset) != 0:
            # per-sample frequency (Doppler correction, decode_funcube.py:228): its own kernel, run now
            f = np.ascontiguousarray(freqOffset, dtype=np.float64).ravel()
            if len(f) != self.length:
                raise ValueError("operands could not be broadcast together with shapes (%d,) (%d,)" % (self.length, len(f)))
            self._settle()                 # (a container that noted this signal in extend() takes its samples first: comm.py:163 copies)
            self._materialise()
            d = self._device(_C64)
            out = DevArray(d.n, _C64)
            df = DevArray.from_host(f)
            _hip.check(_hip.lib().dd_nco_c64_freqs(d.ptr, out.ptr, d.n, df.ptr, float(self.sampRate), int(offset), None),
                       "dd_nco_c64_freqs")
            self._store(out, copy=False)
            return self
        self._record(("nco", _hip.cycles_q64(freqOffset, self.sampRate), int(offset)))
        return self

    def filter(self, filt):
        '''Apply a filter to the signal (comm.py:80-92)'''
        from . import filters as _f
        if isinstance(filt, _f.filter) and filt._fusable():
            self._record(("fir", filt))
            return self
        self.updateSignal(filt.applyOn(self._op_input()))
        return self

    def bwLim(self, tsampRate, strict=False, uniq="abcd"):
        '''Limit the bandwidth by downsampling (comm.py:94-130)'''
        if self.__sampRate < tsampRate:
            raise ValueError("The target sampling rate must be less than current sampling rate")
        if strict:
            # Fourier-domain resample of the whole chunk == scipy.signal.resample (comm.py:110-116)
            from . import _ops
            num = int(tsampRate * self.length / self.sampRate)
            if self._ops and self._ops[-1][0] == "fm":
                self._record(("resample", num))               # real (FM) data: runs with the chain, batched over a chunk list
            else:
                self._settle()
                self._materialise()
                self._store(_ops.resample_fft(self._device(), num), copy=False)
            self.__sampRate = tsampRate
            self.__len = num
        else:
            jumpIndex = int(self.sampRate / tsampRate)
            offset = 0
            if self.__chunker is not None:
                offset = self.__chunker.get(constants.CHUNK_BWLIM + uniq, 0)
                nextOff = (jumpIndex - (self.length - offset) % jumpIndex) % jumpIndex
                self.__chunker.set(constants.CHUNK_BWLIM + uniq, nextOff)
            if jumpIndex != 1 or offset != 0:                  # x[0::1] is x: nothing to run (and a chunk list stays a chunk list)
                self._record(("decim", jumpIndex, int(offset)))
            self.__sampRate = int(self.sampRate / jumpIndex)
            self.__len = len(range(offset, self.__len, jumpIndex))
        return self

    def resamplePoly(self, resampler):
        '''EXTENSION (no counterpart in the reference): polyphase rational resample through a
        ``resample.polyResampler`` created outside the chunk loop; state is carried chunk to chunk.  The signal's
        rate becomes the resampler's output rate (BASELINE config 3: "polyphase resample to 11.025 kS/s").'''
        if int(self.__sampRate) != resampler.inRate:
            raise TypeError("Signals must have same sampling rate")
        out = resampler.applyOn(self._op_input())
        self.__sampRate = resampler.outRate
        self.updateSignal(out)
        return self

    def funcApply(self, func):
        ''' Applies a function to the signal (comm.py:132-144).  Bound ``demod``
        methods of this package's demodulators stay on the device.'''
        from . import demod_fm as _dfm
        owner = getattr(func, "__self__", None)
        if isinstance(owner, _dfm.demod_fm) and getattr(func, "__name__", "") == "demod":
            self.__len = owner._note_call(self.__len)
            self._record(("fm", owner))
            return self
        self.updateSignal(func(self._op_input()))
        return self

    def _op_input(self):
        """What a foreign operator receives: this package's operators accept device
        arrays; anything else gets the NumPy array like in the reference."""
        self._materialise()
        if self._dev is not None and self._dev.dtype == _IQ8:
            self._dev = _convert(self._dev, _C64)             # raw pairs are an ingest format: operators see complex samples
        return self._dev if self._dev is not None else self._host

    # ------------------------------------------------------------------ container ops
    def extend(self, sig):
        ''' Adds another signal to this one at the tail end (comm.py:146-164).
        Device-side append into a geometrically grown buffer (the reference
        re-concatenates everything each chunk).'''
        if self.length == 0:
            self.__sampRate = sig.sampRate
        if not self.__sampRate == sig.sampRate:
            raise TypeError("Signals must have same sampling rate to be extended")
        self._settle()
        if sig.length == 0:
            return self
        if sig._ops and sig._into is None and sig is not self and sig._dev is not None and sig._dev._base is not None:
            # the other signal is a slice of a device-resident recording whose operations are still pending: take its samples
            # when somebody needs ours (its chunk loop may then run as one chunk-list call).  All bookkeeping is done now.
            # (Chunks uploaded one by one are taken at once, so their buffers do not pile up.)
            if self.__len == 0 and not self._lazy and not self._ops:
                self._host = None
                self._dev = None
                self._cap = None
            self._lazy.append(sig)
            sig._into = self
            self.__len += sig.length
            return self
        self._materialise()
        sig._materialise()
        self._append(sig)
        self.__len = self._phys_len()
        return self

    def _phys_len(self):
        if self._dev is not None:
            return self._dev.n
        return len(self._host) if self._host is not None else 0

    def _append(self, sig):
        """append the (materialised) samples of sig to the device buffer; lengths are the caller's business"""
        other = sig._device()
        if other.dtype == _IQ8:
            other = _convert(other, _C64)
        have = self._phys_len()
        if have == 0:
            mine_dt = other.dtype
        else:
            mine_dt = self._device().dtype
            if mine_dt != other.dtype:
                # mixed precision (e.g. float32 FM output appended to float64): widen
                wide = _F64 if (mine_dt.kind == "f" and other.dtype.kind == "f") else _C64
                if mine_dt != wide:
                    self._dev = _convert(self._dev, wide)
                    self._cap = None
                    mine_dt = wide
                if other.dtype != wide:
                    other = _convert(other, wide)
        need = have + other.n
        if self._cap is None or self._cap.n < need or self._cap.dtype != mine_dt:
            cap = DevArray(max(need * 2, 1024), mine_dt)
            if have:
                _hip.check(_hip.lib().dd_memcpy_d2d(cap.ptr, self._device().ptr, have * mine_dt.itemsize, None), "d2d")
            self._cap = cap
        _hip.check(_hip.lib().dd_memcpy_d2d(self._cap.ptr + have * mine_dt.itemsize, other.ptr,
                                            other.n * mine_dt.itemsize, None), "d2d")
        self._dev = self._cap.view(0, need)
        self._host = None

    def updateSignal(self, sig):
        ''' Updates the signal (comm.py:166-181); copies host arrays like the reference'''
        self._settle()
        self._materialise()
        if isinstance(sig, DevArray):
            self._store(sig, copy=False)
        else:
            a = np.array(sig)
            if a.ndim == 0 or not a.size <= a.shape[0]:
                raise TypeError("The signal array must be 1-D")
            self._store(a, copy=False)
        self.__len = len(sig)
        return self

    # ------------------------------------------------------------------ execution
    def _run_ops(self):
        ops, self._ops = self._ops, []
        from . import _ops
        x = self._device()
        i = 0
        while i < len(ops):
            kind = ops[i][0]
            nco = None
            if kind == "nco" and i + 1 < len(ops) and ops[i + 1][0] == "fir":
                nco = ops[i]
                i += 1
                kind = "fir"
            if kind == "fir":
                filt = ops[i][1]
                j = i + 1
                decim = (1, 0)
                fm = None
                if j < len(ops) and ops[j][0] == "decim":
                    decim = (ops[j][1], ops[j][2])
                    j += 1
                if j < len(ops) and ops[j][0] == "fm":
                    fm = ops[j][1]
                    j += 1
                if x.dtype == _C64 or x.dtype == _IQ8:       # raw u8 pairs are widened inside the fused kernel
                    x = _ops.fused(x, filt, nco, decim, fm)
                    i = j
                    continue
                # real (audio-rate) data: stage by stage in float64
                x = filt.applyOn(x)
                i += 1
                continue
            if x.dtype == _IQ8:
                x = _convert(x, _C64)
            if kind == "nco":
                x = _ops.nco(x, ops[i][1], ops[i][2])
            elif kind == "decim":
                x = _ops.decimate(x, ops[i][1], ops[i][2])
            elif kind == "fm":
                x = ops[i][1]._demod_device(x)
            elif kind == "resample":
                x = _ops.resample_fft(x, ops[i][1])
            i += 1
        self._dev = x
        self._host = None
        self._cap = None


def _convert(d, want):
    want = np.dtype(want)
    if d.dtype == want:
        return d
    lib = _hip.lib()
    out = DevArray(d.n, want)
    if d.dtype == _F32 and want == _F64:
        _hip.check(lib.dd_f32_to_f64(d.ptr, out.ptr, d.n, None), "f32->f64")
    elif d.dtype == _F64 and want == _F32:
        _hip.check(lib.dd_f64_to_f32(d.ptr, out.ptr, d.n, None), "f64->f32")
    elif d.dtype == np.dtype(np.complex64) and want == np.dtype(np.complex128):
        _hip.check(lib.dd_f32_to_f64(d.ptr, out.ptr, 2 * d.n, None), "c64->c128")      # interleaved re, im
    elif d.dtype == _IQ8 and want == np.dtype(np.complex64):
        _hip.check(lib.dd_u8iq_to_c64(d.ptr, out.ptr, d.n, None), "dd_u8iq_to_c64")
    elif d.dtype == np.dtype(np.complex128) and want == np.dtype(np.complex64):
        _hip.check(lib.dd_f64_to_f32(d.ptr, out.ptr, 2 * d.n, None), "c128->c64")
    else:
        out = DevArray.from_host(d.to_host().astype(want))
    return out
