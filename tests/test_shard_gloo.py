"""
Multi-rank path on CPU: world_size-2 ``gloo`` processes run the sharding logic of
directdemod_amd/shard.py (ranges, halo priming, output ownership, all_gather) with an
oracle-backed engine standing in for the HIP chain (the engine interface is the
C-ABI chain's: prime / out_count / process).  Concatenated shard outputs must equal
the one-shot stream result exactly (same float64 arithmetic on both sides).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleEngine:
    """dd_chain semantics restated with the oracle (test double; float64)."""

    def __init__(self, x_global, taps, f, fs, decim, fm=True):
        from oracle import dd_oracle as O
        self.O, self.x, self.taps, self.f, self.fs, self.M, self.fm = O, x_global, taps, f, fs, decim, fm
        self.reset()

    def reset(self):
        self.abs = 0
        self.filt = self.O.FilterState(self.taps)
        self.last = None

    def _run(self, a, n):
        O = self.O
        y = self.filt.applyOn(O.nco(self.x[a:a + n], self.f, self.fs, a))
        off = (-a) % self.M
        y = y[off::self.M]
        self.abs = a + n
        if not self.fm:
            return y
        if len(y) == 0:
            return np.zeros(0)
        ang, self.last = O.fm_demod(y, self.last)
        return ang

    def prime(self, a, n_halo, abs_index):
        self.reset()
        if abs_index == 0:
            return
        if n_halo != abs_index:       # history irrelevant: zeros, like dd_chain_prime
            self.filt.zi = np.zeros(len(self.taps) - 1)
        self._run(abs_index - n_halo, n_halo)

    def process(self, a, out, n):
        r = self._run(a, n)
        out[:len(r)] = r
        return len(r)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from directdemod_amd import shard
    from oracle import dd_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = {}
        for M, K, total in ((1, 255, 6001), (34, 151, 20011), (50, 127, 9000)):
            x = O.grid_c64(O.synth_iq_fm(total, 1e6, 77, f_carrier=20e3))
            taps = O.win_hamming(K)
            eng = OracleEngine(x, taps, 20000.0, 1000000, M)
            ranges = shard.shard_ranges(total, world, M)
            a, b = ranges[rank]
            out = np.zeros(total)
            n = shard.run_shard(eng, lambda g: g, a, b, K, M, out)
            assert n == shard.output_count(a, b, M, True), (n, a, b)
            parts = shard.gather_outputs(torch.from_numpy(out), n, world, dist)
            got = np.concatenate([p.numpy() for p in parts])
            ref, _ = O.audio_chain(lambda s, e: x[s:e], total, 1000000, 20000.0, taps, 1000000 // M if M > 1 else 1000000)
            res[(M, K)] = (got.shape == ref.shape) and float(np.max(np.abs(got - ref))) < 1e-9
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_shard_ranges_and_counts():
    sys.path.insert(0, ROOT)
    from directdemod_amd import shard
    assert shard.shard_ranges(100, 4, 1) == [(0, 25), (25, 50), (50, 75), (75, 100)]
    r = shard.shard_ranges(1000, 3, 34)
    assert r[0][0] == 0 and r[-1][1] == 1000 and all(a % 34 == 0 for a, _ in r)
    assert all(r[i][1] == r[i + 1][0] for i in range(2))
    assert shard.halo_len(255, 1) == 255 and shard.halo_len(151, 34) == 184
    for total, world, M in ((1000, 3, 34), (4096, 8, 1), (777, 2, 50), (5, 8, 1)):
        rr = shard.shard_ranges(total, world, M)
        assert sum(shard.kept_count(a, b, M) for a, b in rr) == len(range(0, total, M))
        assert sum(shard.output_count(a, b, M, True) for a, b in rr) == max(0, len(range(0, total, M)) - 1)


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharded_equals_one_shot():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, res in results:
        assert all(res.values()), (rank, res)
