"""
NOAA APT sync detection on the device -- the build's driver for BASELINE config 4
(SURVEY.md 8a rows A1/X1/X2/P).  Stage order and constants follow the reference's
callers (decode_noaa.__audio :600-629, __getAM :631-657, __correlate :659-675,
__correlateAndFindPeaks :677-767, getCrudeSync :769-806, getAccurateSync :808-880);
the arithmetic runs in the HIP kernels behind this package's drop-in classes.
"""
import functools

import numpy as np

from . import _ops, chunker, comm, constants, demod_am, demod_fm, filters
from ._hip import DevArray


def sync_needle(sync_bits, samp_rate, positive=True):
    """decode_noaa.py:689-694"""
    rep = round(samp_rate * constants.NOAA_T)
    if positive:
        return ((np.repeat(sync_bits, rep) * 233) + 11) / 255
    return np.repeat(sync_bits, rep) - 0.5


@functools.lru_cache(maxsize=8)
def _needles(syncs, samp_rate):
    """the needles of sync_needle for a tuple of sync words, stacked (float64, C order, read-only: one array per (words, rate))"""
    a = np.ascontiguousarray(np.stack([sync_needle(np.array(sy), samp_rate) for sy in syncs]), dtype=np.float64)
    a.setflags(write=False)
    return a


@functools.lru_cache(maxsize=1)
def _window_taps():
    """taps of the accurate-sync windows' two zero-phase filters (decode_noaa.py:852, :677)"""
    bh = np.ascontiguousarray(filters.blackmanHarris(151, zeroPhase=True).getB, dtype=np.float64)
    pre = np.ascontiguousarray(filters.hamming(492, zeroPhase=True).getB, dtype=np.float64)
    bh.setflags(write=False)
    pre.setflags(write=False)
    return bh, pre


_prepared = set()           # (device, kind, length) dd_noaa_prepare has been started for in this process


class noaa_sync:
    '''Sync detection of a NOAA APT recording (crude @ ~60 kS/s, accurate @ IQ rate)'''

    def __init__(self, sigsrc, offset, bw=None):
        self.__sigsrc = sigsrc
        self.__offset = offset
        self.__bw = constants.NOAA_FMBW if bw is None else bw
        self.__syncA = None
        self.__syncB = None
        self.__rate = None
        self.__useful = 0
        # the crude sync's envelope stage needs the Hilbert-kernel spectra of its block lengths and the accurate sync that of its window
        # length (closed forms + host transforms, ~25 ms): built on a thread of its own from now on, beside the runtime's first copy, the
        # upload of the recording and the audio chain (dd_noaa_prepare)
        self.__prep = None
        self.__prep_acc = None
        try:
            dec = int(sigsrc.sampFreq / self.__bw)                        # bwLim(NOAA_FMBW): 34 at 2.048 MS/s
            rate1 = int(sigsrc.sampFreq / dec)
            if dec >= 1 and rate1 >= constants.NOAA_CRUDESYNCSAMPRATE and int(rate1 / constants.NOAA_CRUDESYNCSAMPRATE) == 1:
                n_audio = -(-int(sigsrc.length) // dec) - 1               # kept samples of the whole stream, one angle fewer (demod_fm.py:43-49)
                if n_audio > 1:
                    import threading
                    from . import _hip
                    width = int(3 * constants.NOAA_T * len(constants.NOAA_SYNCA) * sigsrc.sampFreq)      # getAccurateSync's windows (:823-825)
                    # (two threads: getCrudeSync waits for the first only; joined there / by getAccurateSync, or by the interpreter at exit.
                    #  Lengths this process has prepared before -- a second decoder object on a recording of the same length -- start none:
                    #  starting and joining a thread is 0.1 ms, more than the warm calls' own host work)
                    dev = _hip.current_device()
                    if (dev, "audio", n_audio) not in _prepared:
                        _prepared.add((dev, "audio", n_audio))
                        self.__prep = threading.Thread(target=_hip.on_callers_device(_ops.noaa_prepare), args=(n_audio, 60000 * 4, 0), daemon=False)
                        self.__prep.start()
                    if (dev, "window", 2 * width) not in _prepared:
                        _prepared.add((dev, "window", 2 * width))
                        self.__prep_acc = threading.Thread(target=_hip.on_callers_device(_ops.noaa_prepare), args=(0, 60000 * 4, 2 * width), daemon=False)
                        self.__prep_acc.start()
        except Exception:
            self.__prep = None

    # ---- FM audio in chunks: one fused kernel per chunk (decode_noaa.py:600-629)
    def audio(self, audioFreq=constants.NOAA_CRUDESYNCSAMPRATE, strictness=False, chunkSize=constants.PROC_CHUNKSIZE):
        src = self.__sigsrc
        audioOut = comm.commSignal(audioFreq)
        bhFilter = filters.blackmanHarris(151)
        fmDemodulator = demod_fm.demod_fm()
        chunkerObj = chunker.chunker(src, chunkSize)
        read = src.read_device if hasattr(src, "read_device") else src.read
        if hasattr(src, "read_device_raw") and src.read_device_raw(0, 1) is not None:
            read = src.read_device_raw          # the recording stays in HBM as raw pairs; the fused kernel widens them
        for a, b in chunkerObj.getChunks:
            sig = comm.commSignal(src.sampFreq, read(a, b), chunkerObj).offsetFreq(self.__offset) \
                .filter(bhFilter).bwLim(self.__bw, uniq="First").funcApply(fmDemodulator.demod) \
                .bwLim(audioFreq, strictness)
            audioOut.extend(sig)
        return audioOut

    # ---- envelope in 240 000-sample blocks (decode_noaa.py:631-657)
    def envelope(self, sig):
        am = demod_am.demod_am().demod_blocks(sig.device_signal, 60000 * 4)
        return comm.commSignal(sig.sampRate, am)

    # ---- normalised correlation + peak pick (decode_noaa.py:677-767)
    def correlate_and_find_peaks(self, sig, sync, use_filter=False, extra=False):
        needle = sync_needle(sync, sig.sampRate)
        hay = sig.device_signal
        if hay.dtype != np.dtype(np.float64):
            hay = comm._convert(hay, np.float64)
        if use_filter:      # zero-phase Hamming(492) pre-filter (default argument at decode_noaa.py:677)
            hay = filters.hamming(492, zeroPhase=True).applyOn(hay)
        cor = _ops.xcorr_norm(hay, needle)
        peaks = _ops.find_peaks(cor, sig.sampRate, len(needle))
        if not extra:
            return peaks
        n = len(needle)
        s = np.asarray(sig.signal, dtype=np.float64)
        corh = cor.to_host()
        heights, tsync = [], []
        for i in peaks:                                          # :754-762
            tsync.append(float(np.average(s[i + n:i + 2 * n])) if i + 2 * n < len(s) else None)
            heights.append(float(corh[i + int(n / 2)]))
        return peaks, heights, tsync

    def getCrudeSync(self, fused=True):
        """decode_noaa.py:769-806.  fused: the audio-rate tail (envelope, both correlations, both peak picks) as ONE device call
        (dd_noaa_crude_tail); False: stage by stage through envelope() / correlate_and_find_peaks() like the reference's code
        (the index lists are the same: tests/test_gpu_audio.py)."""
        if self.__syncA is None or self.__syncB is None:
            audio = self.audio(constants.NOAA_CRUDESYNCSAMPRATE, False)
            res = None
            if self.__prep is not None:
                self.__prep.join()
                self.__prep = None
            if fused:
                res = _ops.crude_tail(audio.device_signal, audio.sampRate,
                                      [sync_needle(constants.NOAA_SYNCA, audio.sampRate), sync_needle(constants.NOAA_SYNCB, audio.sampRate)])
            if res is not None:
                self.__rate = audio.sampRate
                self.__syncA, self.__syncB = res[0]
            else:
                sig = self.envelope(audio)
                self.__rate = sig.sampRate
                self.__syncA = self.correlate_and_find_peaks(sig, constants.NOAA_SYNCA)
                self.__syncB = self.correlate_and_find_peaks(sig, constants.NOAA_SYNCB)

            def _min_dev(s):                                      # :794-801 (the reference's loop over windows, as one array expression)
                d = np.abs(np.diff(s) - (self.__rate * 0.5))
                m = len(d) - constants.NOAA_DETECTCONSSYNCSNUM + 1
                if m <= 0:
                    return np.inf
                return np.min(np.max(np.lib.stride_tricks.sliding_window_view(d, constants.NOAA_DETECTCONSSYNCSNUM), axis=1))
            if _min_dev(self.__syncA) < constants.NOAA_DETECTMAXCHANGE or \
                    _min_dev(self.__syncB) < constants.NOAA_DETECTMAXCHANGE:
                self.__useful = 1
        return [self.__syncA, self.__syncB]

    @property
    def useful(self):
        if self.__syncA is None:
            self.getCrudeSync()
        return self.__useful

    @property
    def crudeRate(self):
        return self.__rate

    def accurate_window(self, startI, endI, sync):
        """one search window of getAccurateSync (decode_noaa.py:852-853)"""
        src = self.__sigsrc
        read = src.read_device if hasattr(src, "read_device") else src.read
        sig = comm.commSignal(src.sampFreq, read(startI, endI)).offsetFreq(self.__offset) \
            .filter(filters.blackmanHarris(151, zeroPhase=True)) \
            .funcApply(demod_fm.demod_fm().demod).funcApply(demod_am.demod_am().demod)
        pk, ht, ts = self.correlate_and_find_peaks(sig, sync, use_filter=True, extra=True)
        return int(pk[0]) + startI, ht[0], ts[0]

    def _gather_windows(self, starts, length):
        """the raw uint8 pairs of the windows, packed [windows][length][2] for one upload"""
        src = self.__sigsrc
        raw = np.empty(len(starts) * length * 2, dtype=np.uint8)
        for w, a in enumerate(starts):
            src.read_raw_u8_into(raw[2 * w * length:2 * (w + 1) * length], a, a + length)
        return raw

    def accurate_windows(self, starts, length, sync, raw=None, resident=True):
        """All search windows of one sync type in one batched device call (dd_noaa_sync_windows): the chain
        of accurate_window over [windows][samples] arrays.  Returns (indices, heights, times) like the
        per-window loop of decode_noaa.py:828-835.  `sync` may also be a list of sync words with `starts` the list of their
        window lists (dd_noaa_sync_windows_multi: both searches of getAccurateSync in one call); the return value is then
        a list of such triples."""
        import ctypes as C
        from . import _hip
        src = self.__sigsrc
        fs = src.sampFreq
        multi = len(starts) > 0 and isinstance(starts[0], (list, tuple, np.ndarray))
        if multi:
            lists, syncs = [list(s) for s in starts], list(sync)
            starts = [a for lst in lists for a in lst]
            group = np.ascontiguousarray([g for g, lst in enumerate(lists) for _ in lst], dtype=np.int32)
        nw = len(starts)
        if nw == 0:
            return [(np.zeros(0, dtype=np.int64), [], []) for _ in lists] if multi else (np.zeros(0, dtype=np.int64), [], [])
        st = (np.arange(nw, dtype=np.int64) * length)
        res = None
        if raw is None and resident and hasattr(src, "resident_raw"):
            lo = int(min(starts))
            res = src.resident_raw(lo, int(max(starts)) + length)       # the recording is (now) in HBM: no gather, no upload
        if res is not None:
            d_raw = res[0]
            st = np.asarray(starts, dtype=np.int64) - lo + res[1]
        elif isinstance(raw, DevArray):
            d_raw = raw
        else:
            raw = self._gather_windows(starts, length) if raw is None else raw
            d_raw = DevArray.from_host(raw.reshape(-1), dtype=np.uint8)
        bh, pre = _window_taps()
        pk = np.empty(nw, dtype=np.int64)
        ht = np.empty(nw, dtype=np.float64)
        ts = np.empty(nw, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        if multi:
            needle = _needles(tuple(tuple(sy) for sy in syncs), fs)
            _hip.check(_hip.lib().dd_noaa_sync_windows_multi(
                d_raw.ptr, 1, st.ctypes.data_as(C.POINTER(C.c_int64)), group.ctypes.data_as(C.POINTER(C.c_int)), nw, int(length),
                _hip.cycles_q64(self.__offset, fs), bh.ctypes.data_as(dp), len(bh), pre.ctypes.data_as(dp), len(pre),
                needle.ctypes.data_as(dp), needle.shape[1], needle.shape[0],
                float(fs), pk.ctypes.data_as(C.POINTER(C.c_int64)), ht.ctypes.data_as(dp), ts.ctypes.data_as(dp), None),
                "dd_noaa_sync_windows_multi")
        else:
            needle = _needles((tuple(sync),), fs)[0]
            _hip.check(_hip.lib().dd_noaa_sync_windows(
                d_raw.ptr, 1, st.ctypes.data_as(C.POINTER(C.c_int64)), nw, int(length), _hip.cycles_q64(self.__offset, fs),
                bh.ctypes.data_as(dp), len(bh), pre.ctypes.data_as(dp), len(pre), needle.ctypes.data_as(dp), len(needle),
                float(fs), pk.ctypes.data_as(C.POINTER(C.c_int64)), ht.ctypes.data_as(dp), ts.ctypes.data_as(dp), None),
                "dd_noaa_sync_windows")
        if np.any(pk == np.iinfo(np.int64).min):
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")      # pk[0] of an empty pick (:853)
        idx = pk + np.asarray(starts, dtype=np.int64)
        hts, tss = ht.tolist(), [None if v != v else v for v in ts.tolist()]
        if not multi:
            return idx, hts, tss
        out, o = [], 0
        for lst in lists:
            out.append((idx[o:o + len(lst)], hts[o:o + len(lst)], tss[o:o + len(lst)]))
            o += len(lst)
        return out

    def getAccurateSync(self, batched=True, resident=True):
        sa, sb = self.getCrudeSync()
        if self.__prep_acc is not None:
            self.__prep_acc.join()
            self.__prep_acc = None
        src = self.__sigsrc
        width = int(3 * constants.NOAA_T * len(constants.NOAA_SYNCA) * src.sampFreq)      # :823-825
        if not hasattr(src, "read_raw_u8_into"):
            batched = False         # a foreign source object (only .read): window by window through the drop-in classes
        out, jobs = [], []
        for crude, sync in ((sa, constants.NOAA_SYNCA), (sb, constants.NOAA_SYNCB)):
            ci = (np.asarray(crude, dtype=np.float64) / self.__rate * src.sampFreq).astype(np.int64)   # int(c) of :829 (c >= 0: truncation)
            keep = (ci - width >= 0) & (ci + width <= src.length)                          # :830-835
            starts = (ci[keep] - width).tolist()
            if batched:
                jobs.append((len(out), starts, sync))
                out.append(None)
                continue
            idx, pks, tms = [], [], []
            for startI in starts:
                i, h, t = self.accurate_window(startI, startI + 2 * width, sync)
                idx.append(i)
                pks.append(h)
                tms.append(t)
            out.append((np.array(idx, dtype=np.int64), pks, tms))
        if jobs and resident and hasattr(src, "resident_raw") and src.read_device_raw(0, 1) is not None:
            # the recording sits in HBM as raw pairs (the crude pass put it there): the windows are read in place, both
            # searches in one device call
            res = self.accurate_windows([st for _, st, _ in jobs], 2 * width, [sync for _, _, sync in jobs])
            for (slot, _, _), (a, b, c) in zip(jobs, res):
                out[slot] = (np.asarray(a, dtype=np.int64), b, c)
            jobs = []
        if jobs:
            # batches of 64 windows (the device batch); the host-side gather of the next batch and its upload
            # (on a side stream) run on a worker thread while the device works on the current one
            import ctypes as C
            from concurrent.futures import ThreadPoolExecutor
            from . import _hip
            up = _hip.stream_create()                      # known to the buffer pool until destroyed

            def feed(st):
                if not st:
                    return None
                return DevArray.from_host(self._gather_windows(st, 2 * width), dtype=np.uint8, stream=up)
            feed = _hip.on_callers_device(feed)            # (the worker thread: this thread's GPU, not HIP's per-thread default)
            parts = [(slot, st[i:i + 64], sync) for slot, st, sync in jobs for i in range(0, max(1, len(st)), 64)]
            res = {slot: ([], [], []) for slot, _, _ in jobs}
            with ThreadPoolExecutor(max_workers=1) as ex:
                nxt = ex.submit(feed, parts[0][1])
                for k, (slot, st, sync) in enumerate(parts):
                    raw = nxt.result()
                    if k + 1 < len(parts):
                        nxt = ex.submit(feed, parts[k + 1][1])
                    a, b, c = self.accurate_windows(st, 2 * width, sync, raw=raw)
                    res[slot][0].extend(a.tolist())
                    res[slot][1].extend(b)
                    res[slot][2].extend(c)
            _hip.stream_destroy(up)
            for slot, _, _ in jobs:
                out[slot] = (np.array(res[slot][0], dtype=np.int64), res[slot][1], res[slot][2])
        return out
