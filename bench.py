#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the DirectDemod hot path on MI355X.

Metric (BASELINE.json): IQ MSamples/s through 255-tap FIR + FM demod; % of HBM
roofline.  Workload = BASELINE.json configs[1] ("C2"):
  synthetic 2.4 MS/s complex64 IQ on the 8-bit source grid, device resident,
  N = 2^26 samples per GPU, commSignal.offsetFreq(25 kHz) NCO + filters.hamming(255)
  + demod_fm.demod, one chunk per step, output float32 radians (N-1 values).
A "step" = one pass of the fused hot path over the GPU's shard.  With --gpus N>1
(launched by torch.distributed.run, one rank per GPU) the stream is N shards of
2^26 samples; rank r re-filters the 256 samples before its shard as a lead-in of
the same launch (absolute-index state, no halo exchange) and no collective sits on
the data path (weak scaling).  `--gather` additionally times an RCCL all_gather of the
decoded output and reports it in "extra" (never in `value`).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_SAMPLE = 12.0        # 8 B complex64 read + 4 B float32 written
FS = 2400000
F_OFFSET = 25000.0
NTAPS = 255


def make_input(torch, n, start, device, seed):
    """Input B of SURVEY.md 8(d): FM tone + noise rounded to the u8 source grid,
    generated on the device in float64 phase (synthetic)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n, 2), dtype=torch.float32, device=device)
    blk = 1 << 22
    for s0 in range(0, n, blk):
        s1 = min(n, s0 + blk)
        t = (torch.arange(s0 + start, s1 + start, device=device, dtype=torch.float64)) / FS
        ph = 2 * np.pi * 25e3 * t + 5.0 * torch.sin(2 * np.pi * 1e3 * t)
        ph = torch.remainder(ph, 2 * np.pi).to(torch.float32)
        re = 60.0 * torch.cos(ph) + 4.0 * torch.randn(s1 - s0, device=device, generator=g)
        im = 60.0 * torch.sin(ph) + 4.0 * torch.randn(s1 - s0, device=device, generator=g)
        out[s0:s1, 0] = torch.clamp(torch.round(re + 127.5), 0, 255) - 127.5
        out[s0:s1, 1] = torch.clamp(torch.round(im + 127.5), 0, 255) - 127.5
    return out


def cpu_baseline(n):
    """Reference CPU path restated (oracle; kind 'port'): np.exp NCO multiply,
    FIR with carried state, np.angle discriminator, one thread."""
    from oracle import dd_oracle as O
    x = O.grid_c64(O.synth_iq_fm(n, FS, 1235))
    taps = O.win_hamming(NTAPS)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        y = O.FilterState(taps).applyOn(O.nco(x, F_OFFSET, FS, 0))
        a, _ = O.fm_demod(y, None)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert len(a) == n - 1
    return {"value": round(n / best / 1e6, 3), "unit": "MSamples/s", "cores": 1, "kind": "port",
            "sample": "2^%d samples of the same workload (numpy float64, best of 2, %.2f s)" % (int(np.log2(n)), best),
            "host_cpus": os.cpu_count()}


def cpu_baseline_all_cores(log2_per_worker=22, timeout_s=120):
    """The same CPU path on every host core the process may use: one process per contiguous shard
    (tools/cpu_allcores.py, a child process that never touches the GPU), bounded by a timeout."""
    import signal
    import subprocess
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "cpu_allcores.py")
    try:
        p = subprocess.Popen([sys.executable, tool, str(log2_per_worker)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             start_new_session=True, text=True)
        try:
            out, _ = p.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)          # the child's own process group (its worker pool included)
            p.communicate()
            return {"error": "timed out after %d s" % timeout_s}
        if p.returncode != 0:
            return {"error": "exit code %d" % p.returncode}
        return json.loads(out.strip().splitlines()[-1])
    except Exception as e:                            # a reported baseline, never a reason to lose the bench line
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp-ms", type=float, default=250.0,
                    help="untimed pre-roll of the same step before the warmup, so the measurement sees the clock the "
                         "chip holds under sustained load (a cold MI355X runs its first ~10 ms of kernels 15-20 %% slower)")
    ap.add_argument("--log2n", type=int, default=26, help="samples per GPU = 2^log2n")
    ap.add_argument("--gather", action="store_true", help="also time an RCCL all_gather of the decoded output")
    ap.add_argument("--force-direct", action="store_true", help="f32 direct-form kernel instead of the MFMA path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log2n", type=int, default=24)
    ap.add_argument("--simulate-rank", type=int, default=None,
                    help="debug: run this rank's shard (halo priming path) on one GPU without torch.distributed")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.simulate_rank is not None:
        rank = args.simulate_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no GPU visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()
    from directdemod_amd import _hip
    _hip.require_gpu()
    lib = _hip.lib()
    _hip.check(lib.dd_set_device(local_rank), "dd_set_device")

    n = 1 << args.log2n
    halo = 256                             # >= ntaps-1+decim (255) and a multiple of 2 samples: keeps the shard 16-byte aligned
    start = rank * n                       # absolute index of this rank's first sample
    pre = halo if rank > 0 else 0
    xin = make_input(torch, n + pre, start - pre, device, 1235 + rank)
    out = torch.empty(n + pre, dtype=torch.float32, device=device)
    first = max(0, pre - 1)                    # index of the shard's first output in `out` (ranks > 0: after the lead-in's)
    torch.cuda.synchronize()

    taps = np.ascontiguousarray(0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(NTAPS) / (NTAPS - 1)))
    flags = _hip.DD_CHAIN_NCO | _hip.DD_CHAIN_FM | (_hip.DD_CHAIN_FORCE_DIRECT if args.force_direct else 0)
    h = C.c_void_p()
    _hip.check(lib.dd_chain_create(C.byref(h), taps.ctypes.data_as(C.POINTER(C.c_double)), NTAPS,
                                   _hip.cycles_q64(F_OFFSET, FS), 1, flags), "dd_chain_create")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n_out = C.c_int64(0)

    def step():
        # new stream position, state derived from the absolute index alone.  Rank 0: the stream start
        # (history of ones).  Rank r: its shard plus the 256-sample lead-in in front of it as one chunk
        # from a zero history -- the lead-in's outputs (the first pre-1, with the filter still filling)
        # are not part of the shard; out[pre-1 : pre-1+n] is what rank r contributes (SURVEY.md 8e:
        # the halo is re-filtered locally, no exchange, no collective).  One launch per step on every rank.
        _hip.check(lib.dd_chain_seek(h, start - pre, stream), "dd_chain_seek")
        _hip.check(lib.dd_chain_process(h, xin.data_ptr(), out.data_ptr(), n + pre, C.byref(n_out), stream), "dd_chain_process")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ramp_steps = 0
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:          # untimed clock pre-roll
        for _ in range(25):
            step()
        torch.cuda.synchronize()
        ramp_steps += 25
    for _ in range(args.warmup):
        step()
    barrier()
    # the hot kernel's launch duration: HIP events on the launch stream around the K timed launches
    # (the chain is ONE kernel launch per step -- edge tiles ride along in it -- so elapsed / K is the
    # average launch duration, launch gaps included)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps
    path = lib.dd_chain_path(h)

    tmax = torch.tensor([dt, kern_ms], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max, kern_ms_max = float(tmax[0]), float(tmax[1])

    extra = {}
    if args.gather and world > 1:
        shard_out = out[first:first + n]                    # this rank's n outputs (a view)
        bufs = [torch.empty_like(shard_out) for _ in range(world)]
        for _ in range(2):
            dist.all_gather(bufs, shard_out)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
            dist.all_gather(bufs, shard_out)
        barrier()
        dtg = time.perf_counter() - t0
        tg = torch.tensor([dtg], dtype=torch.float64, device=device)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        extra["with_all_gather_MSamples_per_s"] = round(world * n * args.steps / float(tg[0]) / 1e6, 1)

    # sanity: the output is a demodulated 1 kHz tone of deviation 5 rad * 2 pi * 1 kHz / fs
    chk = out[first + 1000:first + 1000 + 4096].double().cpu().numpy()
    extra["output_rms_rad"] = float(np.sqrt(np.mean(chk ** 2)))
    extra["clock_preroll"] = {"ms": args.ramp_ms, "steps": ramp_steps}

    if rank == 0 or args.simulate_rank is not None:
        total = world * n * args.steps
        value = total / dt_max / 1e6
        achieved = BYTES_PER_SAMPLE * n / (kern_ms_max * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("bytes_per_launch_log2n_%d" % args.log2n)
            except Exception:
                traffic = None
        res = {
            "metric": "IQ MSamples/s through 255-tap FIR+FM demod",
            "value": round(value, 1),
            "unit": "MSamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C2: 2.4 MS/s complex64 IQ (u8 grid, FM tone + noise), offsetFreq 25 kHz NCO + "
                                   "255-tap Hamming FIR + FM demod, single chunk, 2^%d samples per GPU, device resident"
                                   % args.log2n,
                       "samples_per_gpu": n, "ntaps": NTAPS, "decimation": 1,
                       "kernel_path": {0: "direct-f32", 1: "mfma-f16x3"}.get(path, str(path)),
                       "sharding": "contiguous sample ranges, absolute-index state, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_ms": round(kern_ms_max, 4),
                         "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * n},
            "extra": extra,
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(1 << args.cpu_log2n)
            res["cpu_baseline"]["all_cores"] = cpu_baseline_all_cores()
        print(json.dumps(res))
    lib.dd_chain_destroy(h)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
