"""
bench.py's multi-rank path with the REAL engine on a 1-GPU box: `python bench.py --gpus 2` with
DD_BENCH_ONE_DEVICE=1 starts two ranks under torch.distributed.run that both compute on cuda:0 and talk over
gloo.  Exercised: the launcher, the per-rank build lock, rank 1's 256-sample lead-in (absolute-index state, one
launch per step), the barriers / max-over-ranks reduction and the gather of the decoded stream (rank 0 owns one
output fewer, quirk Q3).  A functional check -- two processes share one GPU, so its rates mean nothing.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(150)
def test_two_ranks_on_one_gpu_through_the_launcher():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DD_BENCH_STUB"):
        env.pop(k, None)
    env["DD_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log2n", "22", "--steps", "3",
                        "--warmup", "1", "--ramp-ms", "5", "--steady-ms", "50"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=120)
    sys.stderr.write("".join(ln + "\n" for ln in r.stderr.splitlines() if ln.startswith("[bench]")))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    n = 1 << 22
    assert j["n_gpus"] == 2 and j["config"]["samples_per_gpu"] == n and j["scaling"] == "weak"
    assert j["config"]["kernel"] == "k_chain_cos1k" and j["data"].startswith("synthetic")
    assert len(j["extra"]["kernel_ms_per_rank"]) == 2 and all(t > 0 for t in j["extra"]["kernel_ms_per_rank"])
    # the gather leg runs without being asked for
    assert j["extra"]["gathered_outputs"] == 2 * n - 1
    assert j["extra"]["with_all_gather_MSamples_per_s"] > 0 and j["extra"]["all_gather_ms_per_step"] > 0
    assert j["extra"]["world_size_seen"] == 2 and j["extra"]["backend"] == "gloo"     # (one-device check: gloo; nccl on a real node)
    assert 0.005 < j["extra"]["output_rms_rad"] < 0.02            # the demodulated 1 kHz tone (deviation 5 rad)
    assert "cpu_baseline" not in j and "side" not in j["extra"]


@pytest.mark.timeout(180)
def test_eight_ranks_on_one_gpu_through_the_launcher():
    """the first real 8-GPU run must not be the first time eight ranks meet.  `bench.py --gpus 8` with
    DD_BENCH_ONE_DEVICE=1: eight ranks on cuda:0 over gloo at 2^22 samples each -- rendezvous on 127.0.0.1, the file-locked build, seven
    256-sample lead-ins (ranks 1..7 start inside the stream: absolute-index state), the max-over-ranks reduction and a gathered stream
    of 8 n - 1 angles (only rank 0 lacks a previous sample, quirk Q3).  Rates mean nothing here."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DD_BENCH_STUB"):
        env.pop(k, None)
    env["DD_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--log2n", "22", "--steps", "3",
                        "--warmup", "1", "--ramp-ms", "5", "--steady-ms", "20"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=150)
    sys.stderr.write("".join(ln + "\n" for ln in r.stderr.splitlines() if ln.startswith("[bench]")))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    n = 1 << 22
    assert j["n_gpus"] == 8 and j["config"]["samples_per_gpu"] == n and j["scaling"] == "weak"
    assert len(j["extra"]["kernel_ms_per_rank"]) == 8 and all(t > 0 for t in j["extra"]["kernel_ms_per_rank"])
    assert j["extra"]["gathered_outputs"] == 8 * n - 1
    assert j["extra"]["world_size_seen"] == 8 and j["extra"]["backend"] == "gloo"
    assert 0.005 < j["extra"]["output_rms_rad"] < 0.02
    assert j["value"] > 0 and j["roofline"]["frac"] > 0
