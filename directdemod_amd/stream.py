"""
Streaming ingest: a source feeding pinned-host ring buffers whose host-to-device
copies run on a side stream, overlapped with the fused kernel of the previous chunk
(BASELINE north_star).  The raw interleaved uint8 I,Q pairs cross
PCIe (2 B/sample instead of 8 for complex64) and the fused kernel widens them itself
(DD_CHAIN_U8_INPUT), so nothing but the decoded output is ever written back.

    ring slot k:  [pinned host u8]  --hipMemcpyAsync (copy stream)-->  [device u8]
                   event "copied[k]"  ->  compute stream waits  ->  dd_chain_process
                   event "consumed[k]" -> copy stream waits before reusing the slot

The chunk loop itself is the reference's (decode_noaa.py:619-624): chunks in order,
state carried inside the chain handle on the device; the host never synchronises
inside the loop (only when the ring wraps onto a slot still being copied from).
"""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import mmap

import numpy as np

from . import _hip, chunker, constants
from ._hip import DevArray, check, lib


class PinnedRing:
    """depth pinned host buffers + matching device buffers, with copy/consume events"""

    def __init__(self, slot_bytes, depth=3, pinned=True):
        _hip.require_gpu()
        self.depth = depth
        self.slot_bytes = int(slot_bytes)
        self.host, self.dev, self.copied, self.consumed = [], [], [], []
        for _ in range(depth):
            if pinned:
                h = C.c_void_p()
                check(lib().dd_host_alloc_pinned(C.byref(h), self.slot_bytes), "dd_host_alloc_pinned")
                self.host.append(h)
            self.dev.append(DevArray(self.slot_bytes, np.uint8))
            for lst in (self.copied, self.consumed):
                e = C.c_void_p()
                check(lib().dd_event_create(C.byref(e)), "dd_event_create")
                lst.append(e)
        self.copy_stream = _hip.stream_create()          # registered with the buffer pool until close()
        self._used = [False] * depth

    def host_view(self, k, nbytes):
        return np.ctypeslib.as_array(C.cast(self.host[k], C.POINTER(C.c_uint8)), shape=(nbytes,))

    def close(self):
        for h in self.host:
            lib().dd_host_free_pinned(h)
        for e in self.copied + self.consumed:
            lib().dd_event_destroy(e)
        _hip.stream_destroy(self.copy_stream)
        for d in self.dev:
            d.free()
        self.host, self.dev = [], []


def _stage(dst, srcarr, pool, nthreads):
    """source array -> pinned slot.  One memcpy thread moves ~10 GB/s, a fifth of what PCIe gen5 x16
    takes; NumPy releases the GIL while copying, so the slot is filled in `nthreads` slices."""
    flat = np.asarray(srcarr).reshape(-1)
    n = flat.size
    if nthreads <= 1 or n < (1 << 22):
        dst[:n] = flat
        return
    step = -(-n // nthreads)
    list(pool.map(lambda i: dst.__setitem__(slice(i, min(n, i + step)), flat[i:min(n, i + step)]), range(0, n, step)))


def stream_fm_chain(src, taps, freq_hz, decim, chunk_size=constants.PROC_CHUNKSIZE, depth=3, compute_stream=None,
                    copy_threads=4, staging="direct"):
    """offsetFreq -> FIR -> decimate -> FM over a u8 source, chunk by chunk, with the
    ingest overlapped.  Returns (device float32 array of all outputs, output rate).

    staging:
      "direct"     -- the chunk's raw pairs go from the source's own memory (array or memmap view, no host copy) to the
                      device slot by hipMemcpyAsync on the copy stream (the runtime stages pageable memory itself);
      "registered" -- the same, but the chunk's pages are pinned IN PLACE first (hipHostRegister, a window of `depth`
                      chunks in flight, unpinned when the slot is reused): a true asynchronous DMA out of the
                      recording's own memory, no staging copy anywhere; falls back to "direct" for ranges that cannot
                      be pinned (e.g. a read-only file mapping on some kernels);
      "pinned"     -- `copy_threads` memcpy threads fill a pinned slot first (for sources without `raw_view`)."""
    from .shard import HipChainEngine
    fs = int(src.sampFreq)
    eng = HipChainEngine(taps, freq_hz, fs, decim, fm=True, nco=True, u8=True, stream=compute_stream)
    ck = chunker.chunker(src, chunk_size)
    chunks = ck.getChunks
    maxlen = max(b - a for a, b in chunks)
    direct = staging in ("direct", "registered") and hasattr(src, "raw_view")
    # (chunks of at least a few pages: then a pinned range holds nothing younger than the head of the NEXT chunk,
    # which is what the unregister rule below relies on; pinning pays off for megabyte chunks only anyway)
    register = direct and staging == "registered" and 2 * min(b - a for a, b in chunks) >= 65536
    registered = [None] * depth                     # host address pinned for the chunk in slot k (page aligned)
    reg_hi = 0                                      # end of the last pinned range
    buf_end = 0                                     # end of the recording in host memory (pinned ranges stop at its last page)
    if direct and register and chunks:
        vall = src.raw_view(chunks[0][0], chunks[-1][1])
        buf_end = vall.ctypes.data + vall.nbytes
    ring = PinnedRing(2 * maxlen, depth, pinned=not direct)
    total_out = max(1, len(range(0, src.length, decim)))
    out = DevArray(total_out, np.float32)
    n_done = 0
    L = lib()
    pool = ThreadPoolExecutor(max(1, copy_threads))
    _hip.register_stream(compute_stream)
    try:
        for i, (a, b) in enumerate(chunks):
            k = i % depth
            n = b - a
            if ring._used[k]:
                check(L.dd_event_sync(ring.consumed[k]), "dd_event_sync")       # slot free again?
            if direct:
                v = src.raw_view(a, b)                                                        # no host copy
                if register:
                    if registered[k] is not None:
                        # the slot's previous chunk (i - depth).  Its pinned range also holds the first bytes of chunk
                        # i - depth + 1 (ranges tile whole pages, see below): that copy must be over as well
                        check(L.dd_event_sync(ring.copied[(k + 1) % depth]), "dd_event_sync")
                        L.dd_host_unregister(registered[k])
                        registered[k] = None
                    # pinned ranges tile the recording in whole pages without overlap (a page can be registered once):
                    # this chunk's range starts where the previous one ended
                    pg = mmap.PAGESIZE
                    lo = max(v.ctypes.data & ~(pg - 1), reg_hi)
                    hi = (v.ctypes.data + 2 * n + pg - 1) & ~(pg - 1)
                    if buf_end:
                        hi = min(hi, (buf_end + pg - 1) & ~(pg - 1))                        # never past the page that holds the recording's last byte
                    # bytes of this chunk that lie in the previous chunk's (still pinned) range: copied on their own,
                    # pinned or not -- a copy may not start inside one registration and run on into other memory
                    split = max(0, min(2 * n, reg_hi - v.ctypes.data)) if reg_hi else 0
                    if hi > lo:
                        if L.dd_host_register(lo, hi - lo) == _hip.DD_OK:
                            registered[k] = lo
                            reg_hi = hi
                        else:
                            register = False                                                  # cannot pin this memory: plain "direct" from the next chunk on
                            reg_hi = 0
                    if split:
                        check(L.dd_memcpy_h2d(ring.dev[k].ptr, v.ctypes.data, split, ring.copy_stream), "h2d")
                        if 2 * n > split:
                            check(L.dd_memcpy_h2d(ring.dev[k].ptr + split, v.ctypes.data + split, 2 * n - split, ring.copy_stream), "h2d")
                    else:
                        check(L.dd_memcpy_h2d(ring.dev[k].ptr, v.ctypes.data, 2 * n, ring.copy_stream), "h2d")
                else:
                    check(L.dd_memcpy_h2d(ring.dev[k].ptr, v.ctypes.data, 2 * n, ring.copy_stream), "h2d")
            else:
                _stage(ring.host_view(k, 2 * n), src.read_raw_u8(a, b), pool, copy_threads)      # file/memmap -> pinned
                check(L.dd_memcpy_h2d(ring.dev[k].ptr, ring.host[k], 2 * n, ring.copy_stream), "h2d")
            check(L.dd_event_record(ring.copied[k], ring.copy_stream), "record")
            check(L.dd_stream_wait_event(compute_stream, ring.copied[k]), "wait")
            got = eng.process(ring.dev[k].ptr, out.ptr + 4 * n_done, n)
            check(L.dd_event_record(ring.consumed[k], compute_stream), "record")
            ring._used[k] = True
            n_done += got
        check(L.dd_stream_sync(compute_stream), "sync")
    finally:
        pool.shutdown(wait=True)
        # also on the exception path: nothing may still be copying into, or reading from, a slot when it is freed
        L.dd_stream_sync(ring.copy_stream)
        L.dd_stream_sync(compute_stream)
        for r in registered:
            if r is not None:
                L.dd_host_unregister(r)
        _hip.unregister_stream(compute_stream)
        eng.close()
        ring.close()
    return out.view(0, n_done), int(fs / decim)
