"""
ctypes binding of the C-ABI in include/directdemod_hip.h (libdirectdemod_hip.so,
built in-tree by __graft_entry__.build()).

There is NO CPU fallback: if the shared library is missing, or no MI355X is
visible when a compute entry point is reached, the call raises.
"""
import ctypes as C
import functools
import os
import threading
from fractions import Fraction

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdirectdemod_hip.so")
# DD_LIB_PATH (diagnostics only: tools/ A/B runs of ablation builds, build/variants/lib_N.so) loads another build of the same
# C-ABI INSTEAD of the product library and says so on stderr; the product file is never overwritten by a measurement script
if os.environ.get("DD_LIB_PATH"):
    LIB_PATH = os.path.abspath(os.environ["DD_LIB_PATH"])
    import sys as _sys
    _sys.stderr.write("directdemod_amd: DD_LIB_PATH set, loading %s instead of the product library\n" % LIB_PATH)

DD_OK = 0
DD_ERR_INVALID = -1
DD_ERR_HIP = -2
DD_ERR_NOMEM = -3
DD_ERR_UNSUPPORTED = -4
DD_ERR_NODEVICE = -5
DD_ERR_TIMEOUT = -6

DD_HIST_ZEROS, DD_HIST_ONES, DD_HIST_GIVEN = 0, 1, 2
DD_CHAIN_NCO, DD_CHAIN_FM, DD_CHAIN_U8_INPUT, DD_CHAIN_FORCE_DIRECT, DD_CHAIN_TIGHT = 1, 2, 4, 8, 16
(DD_KERNEL_NONE, DD_KERNEL_DENSE_F32, DD_KERNEL_DECIM_TILES, DD_KERNEL_DECIM_PERSISTENT, DD_KERNEL_MFMA_WS,
 DD_KERNEL_MFMA_TILES, DD_KERNEL_MFMA_AB, DD_KERNEL_FFT_OS, DD_KERNEL_DECIM_MULTI, DD_KERNEL_COS_RS, DD_KERNEL_DECIM_WAVE,
 DD_KERNEL_DECIM_BLOCKS) = range(12)


def decim_wave_kernel(K, M):
    """Which of the two wave-per-row decimating kernels takes (K taps, even decimation M in 8..64): block sums where an output needs at most
    eight of them (dd_decimw.hip, round 6), else one window per lane."""
    return DD_KERNEL_DECIM_BLOCKS if -(-K // M) <= 8 else DD_KERNEL_DECIM_WAVE
# element type of raw interleaved uint8 I,Q pairs held on the device (source.py:117-118 not yet applied): 2 B/sample
IQ8 = np.dtype([("i", np.uint8), ("q", np.uint8)])


class HipError(RuntimeError):
    pass


_p = C.c_void_p
_i64 = C.c_int64
_u64 = C.c_uint64
_int = C.c_int
_sz = C.c_size_t
_pp = C.POINTER(C.c_void_p)
_pi64 = C.POINTER(C.c_int64)

# name -> (restype, argtypes); must list every symbol include/directdemod_hip.h declares
SIGNATURES = {
    "dd_last_error": (C.c_char_p, []),
    "dd_version": (C.c_char_p, []),
    "dd_device_count": (_int, [C.POINTER(_int)]),
    "dd_set_device": (_int, [_int]),
    "dd_get_device": (_int, [C.POINTER(_int)]),
    "dd_copy_warmup": (_int, []),
    "dd_code_warmup": (_int, []),
    "dd_device_name": (_int, [C.c_char_p, _int]),
    "dd_malloc": (_int, [_pp, _sz]),
    "dd_free": (_int, [_p]),
    "dd_memset": (_int, [_p, _int, _sz, _p]),
    "dd_host_alloc_pinned": (_int, [_pp, _sz]),
    "dd_host_free_pinned": (_int, [_p]),
    "dd_host_register": (_int, [_p, _sz]),
    "dd_host_unregister": (_int, [_p]),
    "dd_debug_fill_lds": (_int, [C.c_uint32, _p]),
    "dd_debug_select_kernel": (_int, [C.c_char_p]),
    "dd_debug_seam": (_int, [_int, _int]),
    "dd_debug_fft1k_plan": (_int, [_i64, _int, _int, _int, _int, C.POINTER(_int)]),
    "dd_debug_cos1k_plan": (_int, [_i64, _int, _int, _int, C.POINTER(_int)]),
    "dd_debug_decimw_plan": (_int, [_i64, _i64, _int, _int, _int, _int, C.POINTER(_i64)]),
    "dd_debug_cos_fit": (_int, [C.POINTER(C.c_double), _int, C.POINTER(C.c_double), C.POINTER(_int)]),
    "dd_debug_sync_envelope": (_int, [_p, _i64, _int, _int, _p, _p]),
    "dd_memcpy_h2d": (_int, [_p, _p, _sz, _p]),
    "dd_memcpy_d2h": (_int, [_p, _p, _sz, _p]),
    "dd_memcpy_d2d": (_int, [_p, _p, _sz, _p]),
    "dd_stream_create": (_int, [_pp]),
    "dd_stream_destroy": (_int, [_p]),
    "dd_stream_sync": (_int, [_p]),
    "dd_event_create": (_int, [_pp]),
    "dd_event_destroy": (_int, [_p]),
    "dd_event_record": (_int, [_p, _p]),
    "dd_event_elapsed_ms": (_int, [_p, _p, C.POINTER(C.c_float)]),
    "dd_event_sync": (_int, [_p]),
    "dd_stream_wait_event": (_int, [_p, _p]),
    "dd_u8iq_to_c64": (_int, [_p, _p, _i64, _p]),
    "dd_nco_c64": (_int, [_p, _p, _i64, _u64, _i64, _p]),
    "dd_nco_c64_freqs": (_int, [_p, _p, _i64, _p, C.c_double, _i64, _p]),
    "dd_fir_create": (_int, [_pp, C.POINTER(C.c_double), _int]),
    "dd_fir_destroy": (_int, [_p]),
    "dd_fir_reset": (_int, [_p, _int, _p, _p]),
    "dd_fir_reset_hist_f64": (_int, [_p, C.POINTER(C.c_double), _p]),
    "dd_fir_c64": (_int, [_p, _p, _p, _i64, _int, _p]),
    "dd_fir_f64": (_int, [_p, _p, _p, _i64, _int, _p]),
    "dd_filtfilt_f64": (_int, [C.POINTER(C.c_double), _int, _p, _p, _i64, _int, _p]),
    "dd_filtfilt_c64": (_int, [C.POINTER(C.c_double), _int, _p, _p, _i64, _p]),
    "dd_iir_create": (_int, [_pp, C.POINTER(C.c_double), C.POINTER(C.c_double), _int, C.POINTER(C.c_double)]),
    "dd_iir_destroy": (_int, [_p]),
    "dd_iir_f64": (_int, [_p, _p, _p, _i64, _int, _int, _p]),
    "dd_iir_c64": (_int, [_p, _p, _p, _i64, _int, _p]),
    "dd_iir_filtfilt_f64": (_int, [_p, _p, _p, _i64, _int, _p]),
    "dd_decimate": (_int, [_p, _p, _i64, _int, _int, _int, _pi64, _p]),
    "dd_fm_create": (_int, [_pp]),
    "dd_fm_destroy": (_int, [_p]),
    "dd_fm_reset": (_int, [_p]),
    "dd_fm_discrim_c64": (_int, [_p, _p, _p, _i64, _int, _pi64, _p]),
    "dd_fused_process": (_int, [_p, _p, _p, _p, _i64, _int, _u64, _i64, _int, _int, _int, _int, _pi64, _p]),
    "dd_chain_create": (_int, [_pp, C.POINTER(C.c_double), _int, _u64, _int, _int]),
    "dd_chain_destroy": (_int, [_p]),
    "dd_chain_reset": (_int, [_p, _p]),
    "dd_chain_seek": (_int, [_p, _i64, _p]),
    "dd_chain_prime": (_int, [_p, _p, _i64, _i64, _p]),
    "dd_chain_out_count": (_i64, [_p, _i64]),
    "dd_chain_process": (_int, [_p, _p, _p, _i64, _pi64, _p]),
    "dd_fir_last_kernel": (_int, [_p]),
    "dd_fir_launch_count": (C.c_longlong, [_p]),
    "dd_fused_process_chunks": (_int, [_p, _p, _p, _p, _pi64, _int, _int, C.c_uint64, _i64, _int, _int, _int, _pi64, _p]),
    "dd_chain_process_chunks": (_int, [_p, _p, _p, _pi64, _int, _pi64, _p]),
    "dd_chain_path": (_int, [_p]),
    "dd_chain_last_kernel": (_int, [_p]),
    "dd_resample_fft_f64": (_int, [_p, _p, _i64, _i64, _p]),
    "dd_resample_fft_chunks": (_int, [_p, _int, _pi64, _pi64, _p, _pi64, _pi64, _int, _p]),
    "dd_rpoly_create": (_int, [_pp, C.POINTER(C.c_double), _int, _int, _int, _i64]),
    "dd_rpoly_destroy": (_int, [_p]),
    "dd_rpoly_reset": (_int, [_p]),
    "dd_rpoly_out_count": (_i64, [_p, _i64, _int]),
    "dd_rpoly_process": (_int, [_p, _p, _i64, _int, _p, _pi64, _p]),
    "dd_am_envelope_f64": (_int, [_p, _p, _i64, _i64, _p]),
    "dd_xcorr_norm_f64": (_int, [_p, _i64, C.POINTER(C.c_double), _int, _p, _p]),
    "dd_find_peaks_f64": (_int, [_p, _i64, C.c_double, _int, _pi64, _int, C.POINTER(_int), _p]),
    "dd_noaa_sync_windows": (_int, [_p, _int, _pi64, _int, _i64, C.c_uint64, C.POINTER(C.c_double), _int,
                                    C.POINTER(C.c_double), _int, C.POINTER(C.c_double), _int, C.c_double,
                                    _pi64, C.POINTER(C.c_double), C.POINTER(C.c_double), _p]),
    "dd_noaa_sync_windows_multi": (_int, [_p, _int, _pi64, C.POINTER(_int), _int, _i64, C.c_uint64, C.POINTER(C.c_double), _int,
                                          C.POINTER(C.c_double), _int, C.POINTER(C.c_double), _int, _int, C.c_double,
                                          _pi64, C.POINTER(C.c_double), C.POINTER(C.c_double), _p]),
    "dd_noaa_prepare": (_int, [_i64, _i64, _i64, _p]),
    "dd_noaa_crude_tail": (_int, [_p, _int, _i64, C.c_double, _i64, C.POINTER(C.c_double), _int, _int, _p,
                                  _pi64, _int, C.POINTER(_int), _p]),
    "dd_afsk_binary_filter_f64": (_int, [_p, _i64, C.POINTER(C.c_double), _int, _p, _p]),
    "dd_afsk_edges_f64": (_int, [_p, _i64, _int, _p, _p]),
    "dd_abs_f64": (_int, [_p, _int, _p, _i64, _p]),
    "dd_f32_to_f64": (_int, [_p, _p, _i64, _p]),
    "dd_f64_to_f32": (_int, [_p, _p, _i64, _p]),
}

_lib = None
_lock = threading.Lock()
_gpu_checked = False
_last_path = None


def _bind_to_pytorch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, requested by libtorch_hip as plain "libamdhip64.so"): if this library has
    already pulled in the system runtime when torch initialises, the process ends up with two
    runtimes and torch reports "no ROCm-capable device".  Loading torch's copy first -- without
    importing torch -- makes our DT_NEEDED resolve to it, whatever the import order (bench.py,
    the torch.distributed sharding path and the tests use both in one process).
    DD_HIP_SYSTEM_RUNTIME=1 keeps the system runtime."""
    if os.environ.get("DD_HIP_SYSTEM_RUNTIME"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """dlopen the C-ABI library and bind every declared symbol (no GPU needed)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "directdemod_amd: %s is missing -- run `python __graft_entry__.py` (hipcc, gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        _bind_to_pytorch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return _lib


def lib():
    return _lib if _lib is not None else load()


def last_error():
    return lib().dd_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    if rc == DD_OK:
        return
    msg = "%s: %s (code %d)" % (what or "directdemod_hip", last_error(), rc)
    if rc == DD_ERR_INVALID:
        raise ValueError(msg)
    if rc == DD_ERR_NOMEM:
        raise MemoryError(msg)
    if rc == DD_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise HipError(msg)


def select_kernel(name=None):
    """Tools and tests: force one of the M = 1 chain kernels ("ab", "ws", "fft1k", "cos1k") for every later launch of this process,
    or go back to the choice by tap class (None / "auto").  Mirrors the choice into os.environ["DD_MFMA_KERNEL"] (which only
    seeds the library's choice, once per process) so that code which looks there sees the same thing."""
    check(lib().dd_debug_select_kernel((name or "auto").encode()), "dd_debug_select_kernel")
    if name and name != "auto":
        os.environ["DD_MFMA_KERNEL"] = name
    else:
        os.environ.pop("DD_MFMA_KERNEL", None)


def require_gpu():
    """Fail loudly when the HIP path cannot run."""
    global _gpu_checked
    if _gpu_checked:
        return
    n = _int(0)
    rc = lib().dd_device_count(C.byref(n))
    if rc != DD_OK or n.value < 1:
        raise HipError("directdemod_amd needs an AMD MI355X (gfx950) visible to HIP; none found: %s. "
                       "There is no CPU fallback." % last_error())
    _gpu_checked = True
    _start_copy_warmup()


_warm = None
_warm_copy_done = threading.Event()


def current_device():
    """The calling thread's HIP device ordinal (dd_get_device)."""
    d = _int(0)
    check(lib().dd_get_device(C.byref(d)), "dd_get_device")
    return d.value


def on_callers_device(fn):
    """`fn` wrapped for a helper thread: the thread first takes the device that is current on the thread that calls THIS function (HIP's
    current device is per host thread and starts at 0 -- a helper of rank k of a multi-GPU job would otherwise work on GPU 0)."""
    dev = current_device()

    def run(*a, **kw):
        check(lib().dd_set_device(dev), "dd_set_device")
        return fn(*a, **kw)
    return run


def _start_copy_warmup():
    """The first host-to-device copy of a process costs ~90 ms inside the HIP runtime whatever its size (tools/debug/first_copy.py: a 4 KB
    copy 88-93 ms, the 40 MB after it 8 ms).  A 4 KB copy on a thread of its own, started when the GPU is first touched, takes that out
    of the first real upload: by the time a caller has opened its recording the copy path is up.  (source.read_device_raw waits for it.)"""
    global _warm
    if _warm is not None or os.environ.get("DD_NO_COPY_WARMUP"):
        return

    dev = current_device()

    def run():
        # (HIP keeps the current device per host thread, default 0: take the caller's -- under bench.py --gpus N every rank's main
        #  thread has chosen its own GPU -- and make a plain synchronous copy: no stream of the library is touched, no seam word looked at)
        try:
            if lib().dd_set_device(dev) == DD_OK:
                lib().dd_copy_warmup()
                _warm_copy_done.set()
                if not os.environ.get("DD_NO_CODE_WARMUP"):
                    lib().dd_code_warmup()      # (the code objects: 1-4 ms per translation unit that the first launches would pay)
        except Exception:
            pass
        finally:
            _warm_copy_done.set()
    # (not a daemon: an interpreter that exits right away waits the few milliseconds this takes instead of tearing the HIP runtime
    #  down under a thread that is inside it)
    _warm = threading.Thread(target=run, name="dd-copy-warmup", daemon=False)
    _warm.start()


def wait_copy_warmup():
    """until the warm-up thread's copies are done (its code-object loads may still be running)"""
    if _warm is not None and _warm.is_alive():
        _warm_copy_done.wait()


def device_name():
    require_gpu()
    buf = C.create_string_buffer(256)
    check(lib().dd_device_name(buf, 256), "dd_device_name")
    return buf.value.decode()


@functools.lru_cache(maxsize=256)
def _cycles_q64(freq_hz, samp_rate):
    fr = Fraction(freq_hz) / Fraction(samp_rate)
    q = round(fr * (1 << 64))
    return int(q) % (1 << 64)


def cycles_q64(freq_hz, samp_rate):
    """frac(f/fs) * 2^64 as an unsigned 64-bit integer, from the exact rational value
    of the two floats (the reference forms 2*pi*f*n/fs in float64, comm.py:77).  (A chunk loop asks for the same pair
    once per chunk: the rational arithmetic, 14 us, is remembered.)"""
    return _cycles_q64(float(freq_hz), int(samp_rate))


def set_last_path(p):
    global _last_path
    _last_path = p


def last_chain_path():
    return {None: "none", 0: "direct-f32", 1: "mfma-f16x3"}.get(_last_path, str(_last_path))


# ------------------------------------------------------------------ device arrays
_DT = {
    "c64": np.dtype(np.complex64), "f32": np.dtype(np.float32),
    "f64": np.dtype(np.float64), "c128": np.dtype(np.complex128), "u8": np.dtype(np.uint8),
}


# ------------------------------------------------------------------ device buffer pool
# hipMalloc / hipFree cost tens of microseconds each and hipFree synchronises the device; a chunk
# loop allocates the same few sizes over and over (decode_noaa.py:619-624: one set of intermediates
# per chunk).  Freed buffers are kept by size (rounded up to 4 KiB) and handed out again.
# Stream safety: work on ONE stream is ordered, so a buffer freed and reused on the same stream needs
# nothing.  The package also drives non-blocking side streams (the ring feeder's copy stream, a
# caller's compute stream, the accurate-sync upload stream): while any such stream is registered
# (stream_create / register_stream), a freed buffer is parked together with one event per live stream
# (null stream included), recorded at free time, and whoever takes it out of the pool first waits for
# those events -- so asynchronous work that still touches the buffer on any stream has finished before
# a new owner's copy or kernel on another stream can overwrite it.  Without side streams the events are
# skipped.  At most DD_POOL_BYTES are held (default 16 GiB of the part's 288 GB -- 1 GiB until round 6: two 1 GiB arrays alive in a loop, a complex128
# copy of 2^26 samples and its filtered output, then cost a hipMalloc + hipFree of 15-30 ms on every pass); DD_POOL_BYTES=0 disables the pool.
_POOL_LIMIT = int(os.environ.get("DD_POOL_BYTES", str(16 << 30)))
_pool = {}
_pool_bytes = 0
_pool_lock = threading.Lock()
_streams = {}                 # live side streams: handle value -> use count
_event_spare = []             # recycled hipEvent handles


def stream_create():
    """a non-blocking side stream, known to the buffer pool until stream_destroy"""
    s = C.c_void_p()
    check(lib().dd_stream_create(C.byref(s)), "dd_stream_create")
    register_stream(s)
    return s


def stream_destroy(s):
    unregister_stream(s)
    lib().dd_stream_destroy(s)


def _sval(s):
    return int(s.value or 0) if isinstance(s, C.c_void_p) else int(s or 0)


def register_stream(s):
    """tell the buffer pool about a stream created elsewhere (e.g. a torch stream handed in as compute stream)"""
    v = _sval(s)
    if v:
        with _pool_lock:
            first = not _streams
            _streams[v] = _streams.get(v, 0) + 1
        if first and _lib is not None:
            # buffers freed while only the null stream was in use were parked without a fence (null-stream work may
            # still be pending on them); from now on a parked buffer can be handed to a non-blocking stream, so let
            # that work finish once
            _lib.dd_stream_sync(None)


def unregister_stream(s):
    v = _sval(s)
    with _pool_lock:
        if v in _streams:
            _streams[v] -= 1
            if _streams[v] <= 0:
                del _streams[v]


def _fence_events():
    """events marking 'everything submitted so far' on every live stream; [] when only the null stream is in use"""
    with _pool_lock:
        live = list(_streams)
    if not live:
        return []
    evs = []
    L = lib()
    try:
        for sv in [0] + live:
            with _pool_lock:
                e = _event_spare.pop() if _event_spare else None
            if e is None:
                e = C.c_void_p()
                if L.dd_event_create(C.byref(e)) != DD_OK:
                    raise HipError("dd_event_create: " + last_error())
            evs.append(e)
            if L.dd_event_record(e, C.c_void_p(sv) if sv else None) != DD_OK:
                raise HipError("dd_event_record: " + last_error())
    except Exception:
        with _pool_lock:
            _event_spare.extend(evs)            # (the events made so far go back to the spare list, not lost)
        raise
    return evs


def _fence_wait(evs):
    L = lib()
    for e in evs:
        L.dd_event_sync(e)
    with _pool_lock:
        _event_spare.extend(evs)


def _pool_round(nbytes):
    return (max(16, int(nbytes)) + 4095) & ~4095


def _pool_alloc(nbytes):
    global _pool_bytes
    size = _pool_round(nbytes)
    with _pool_lock:
        lst = _pool.get(size)
        ent = None
        if lst:
            _pool_bytes -= size
            ent = lst.pop()
    if ent is not None:
        if ent[1]:
            _fence_wait(ent[1])
        return ent[0], size
    p = C.c_void_p()
    rc = lib().dd_malloc(C.byref(p), size)
    if rc == DD_ERR_NOMEM and _pool:
        pool_trim(0)
        rc = lib().dd_malloc(C.byref(p), size)
    check(rc, "dd_malloc")
    return p.value, size


def _pool_free(ptr, size):
    global _pool_bytes
    with _pool_lock:
        keep = _pool_bytes + size <= _POOL_LIMIT
    if keep:
        try:
            evs = _fence_events()
        except Exception:
            keep = False
    if keep:
        with _pool_lock:
            _pool.setdefault(size, []).append((ptr, evs))
            _pool_bytes += size
        return
    lib().dd_free(ptr)          # hipFree synchronises the device


def pool_trim(keep_bytes=0):
    """give pooled device memory back to the driver until at most keep_bytes are held"""
    global _pool_bytes
    with _pool_lock:
        for size in sorted(_pool, reverse=True):
            lst = _pool[size]
            while lst and _pool_bytes > keep_bytes:
                ptr, evs = lst.pop()
                _event_spare.extend(evs)
                lib().dd_free(ptr)
                _pool_bytes -= size
        for size in [k for k, v in _pool.items() if not v]:
            del _pool[size]


class DevArray:
    """1-D array resident in HBM (owned hipMalloc buffer or a view of one)."""

    __slots__ = ("ptr", "n", "dtype", "_owner", "_base", "_size")

    def __init__(self, n, dtype, ptr=None, base=None):
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self._base = base
        self._size = 0
        if ptr is None:
            require_gpu()
            self.ptr, self._size = _pool_alloc(self.n * self.dtype.itemsize)
            self._owner = True
        else:
            self.ptr = int(ptr)
            self._owner = False

    def __len__(self):
        return self.n

    @property
    def nbytes(self):
        return self.n * self.dtype.itemsize

    def view(self, start, count):
        return DevArray(count, self.dtype, ptr=self.ptr + start * self.dtype.itemsize,
                        base=self if self._base is None else self._base)

    @staticmethod
    def from_host(a, dtype=None, stream=None):
        a = np.ascontiguousarray(a, dtype=dtype)
        d = DevArray(a.size, a.dtype)
        if a.size:
            check(lib().dd_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes, stream), "h2d")
            check(lib().dd_stream_sync(stream), "sync")
        return d

    def to_host(self, stream=None):
        out = np.empty(self.n, dtype=self.dtype)
        if self.n:
            check(lib().dd_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, stream), "d2h")
            check(lib().dd_stream_sync(stream), "sync")
        return out

    def free(self):
        if getattr(self, "_owner", False) and self.ptr:
            try:
                _pool_free(self.ptr, self._size)
            except Exception:
                pass
            self.ptr = 0
            self._owner = False

    def __del__(self):
        self.free()


def to_device(x, dtype):
    if isinstance(x, DevArray):
        if x.dtype != np.dtype(dtype):
            raise TypeError("device array has dtype %s, expected %s" % (x.dtype, np.dtype(dtype)))
        return x
    return DevArray.from_host(np.asarray(x), dtype=dtype)


def sync(stream=None):
    check(lib().dd_stream_sync(stream), "sync")
