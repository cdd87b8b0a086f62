"""
Chunk-sharded data parallelism for the hot path (the reference has
no parallelism at all -- its chunk loop is strictly sequential because state is
carried chunk to chunk).

Every carried quantity of the NCO -> FIR -> decimate -> FM chain is a pure function
of the absolute sample position: NCO phase = f(n), FIR state = the previous ntaps-1
inputs, FM state = the previous kept FIR output, decimation grid = multiples of M.
So rank r of R takes the contiguous range [start_r, stop_r) of the stream, re-reads
the (ntaps-1+M)-sample halo in front of it from the shared input and *recomputes*
its state (dd_chain_prime): no halo exchange and no collective on the data path.
The decoded outputs can optionally be gathered over RCCL (torch.distributed
backend "nccl"; "gloo" in the CPU tests).

This module is host logic only; the arithmetic is behind an ``engine`` with the
C-ABI chain semantics (HipChainEngine below; the CPU tests substitute a NumPy
stand-in engine to exercise the sharding without a GPU).
"""
import ctypes as C

import numpy as np


def shard_ranges(total, world, decim=1):
    """Contiguous [start, stop) per rank; starts are multiples of ``decim`` so every
    rank's first kept sample is its first sample."""
    per = -(-total // world)
    per = -(-per // decim) * decim
    out = []
    for r in range(world):
        a = min(total, r * per)
        b = min(total, (r + 1) * per)
        out.append((a, b))
    return out


def halo_len(ntaps, decim):
    """samples in front of a shard needed to rebuild its state exactly"""
    return ntaps - 1 + decim


def kept_count(start, stop, decim):
    """kept (decimated) samples with global index in [start, stop)"""
    first = -(-start // decim) * decim
    return 0 if first >= stop else (stop - 1 - first) // decim + 1


def output_count(start, stop, decim, fm):
    """outputs a shard emits: FM pairs (y[k], y[k-1]) belong to the shard that owns k;
    the very first kept sample of the stream has no predecessor (quirk Q3)."""
    n = kept_count(start, stop, decim)
    if fm and start == 0 and n > 0:
        n -= 1
    return n


class HipChainEngine:
    """dd_chain_* behind the engine interface (device pointers in, device pointers out)"""

    def __init__(self, taps, freq_hz, fs, decim, fm=True, nco=True, u8=False, stream=None):
        from . import _hip
        self._hip = _hip
        _hip.require_gpu()
        self.lib = _hip.lib()
        taps = np.ascontiguousarray(taps, dtype=np.float64)
        self.ntaps, self.decim, self.fm = len(taps), decim, fm
        flags = (_hip.DD_CHAIN_NCO if nco else 0) | (_hip.DD_CHAIN_FM if fm else 0) | (_hip.DD_CHAIN_U8_INPUT if u8 else 0)
        self.h = C.c_void_p()
        _hip.check(self.lib.dd_chain_create(C.byref(self.h), taps.ctypes.data_as(C.POINTER(C.c_double)), len(taps),
                                            _hip.cycles_q64(freq_hz, fs), decim, flags), "dd_chain_create")
        self.stream = stream
        self.elem = 2 if u8 else 8

    def prime(self, in_ptr, n_halo, abs_index):
        self._hip.check(self.lib.dd_chain_prime(self.h, in_ptr, n_halo, abs_index, self.stream), "dd_chain_prime")

    def out_count(self, n):
        return int(self.lib.dd_chain_out_count(self.h, n))

    def process(self, in_ptr, out_ptr, n):
        got = C.c_int64(0)
        self._hip.check(self.lib.dd_chain_process(self.h, in_ptr, out_ptr, n, C.byref(got), self.stream), "dd_chain_process")
        return got.value

    def close(self):
        if self.h:
            self.lib.dd_chain_destroy(self.h)
            self.h = None


def run_shard(engine, read_ptr, start, stop, ntaps, decim, out_ptr):
    """Process global samples [start, stop) on this rank.  ``read_ptr(a)`` returns the
    address of global sample ``a`` in a buffer that holds [start - halo, stop)."""
    if stop <= start:
        return 0
    halo = min(start, halo_len(ntaps, decim))
    engine.prime(read_ptr(start - halo) if halo else None, halo, start)
    return engine.process(read_ptr(start), out_ptr, stop - start)


def gather_outputs(local, count, world, dist, device=None, force=False):
    """all_gather of variable-length per-rank outputs (torch tensors); returns the list of
    per-rank tensors trimmed to their counts.  No-op for world == 1 (unless `force`: the
    collectives then run on the one-rank group, a functional check of the backend)."""
    import torch
    if world == 1 and not force:
        return [local[:count]]
    cnt = torch.tensor([count], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    mx = int(max(int(c) for c in counts))
    pad = torch.zeros(mx, dtype=local.dtype, device=local.device)
    pad[:count] = local[:count]
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return [b[:int(c)] for b, c in zip(bufs, counts)]
