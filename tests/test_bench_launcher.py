"""
bench.py's launcher path on CPU: `python bench.py --gpus 2` (no WORLD_SIZE) must start two ranks under
torch.distributed.run, never import torch in the launcher itself, and relay rank 0's single JSON line
with n_gpus = 2.  DD_BENCH_STUB=1 swaps the HIP engine for a do-nothing stand-in and RCCL for gloo
(the line it prints is marked `data: "stub"`): what is tested is the launch / barrier / max-over-ranks /
gather plumbing, not a measurement.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_args, env_extra=None, timeout=240):
    env = dict(os.environ)
    env["DD_BENCH_STUB"] = "1"
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, capture_output=True, text=True,
                       env=env, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


@pytest.mark.timeout(300)
def test_gpus_2_launches_two_ranks_and_prints_one_line():
    r, lines = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--ramp-ms", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 1
    assert j["data"] == "stub" and j["config"]["kernel_path"] == "stub"
    assert j["scaling"] == "weak" and j["higher_is_better"] is True
    assert len(j["extra"]["kernel_ms_per_rank"]) == 2
    # gather leg: rank 0 of the stream contributes n-1 outputs (quirk Q3), every other rank n
    assert j["extra"]["gathered_outputs"] == 2 * j["config"]["samples_per_gpu"] - 1
    assert "with_all_gather_MSamples_per_s" in j["extra"] and "all_gather_ms_per_step" in j["extra"]      # without --gather
    assert j["extra"]["world_size_seen"] == 2 and j["extra"]["backend"] == "gloo"
    assert "cpu_baseline" not in j                      # rank 0 at N = 1 only


@pytest.mark.timeout(300)
def test_no_gather_skips_the_collective_leg():
    r, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--ramp-ms", "2", "--no-gather"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(lines[0])
    assert "with_all_gather_MSamples_per_s" not in j["extra"] and j["extra"]["world_size_seen"] == 2


@pytest.mark.timeout(120)
def test_short_node_gets_one_clear_line():
    """--gpus 4 on a node that shows fewer GPUs (this container shows none): the launcher probes the device count in a
    throw-away child and leaves with one line and a non-zero code, before building or starting any rank"""
    env = dict(os.environ)
    for k in ("DD_BENCH_STUB", "DD_BENCH_ONE_DEVICE", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env, timeout=100, cwd=ROOT)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    msg = [ln for ln in r.stderr.splitlines() if ln.startswith("bench.py:")]
    assert len(msg) == 1 and "--gpus 4" in msg[0] and "visible" in msg[0], r.stderr[-500:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(120)
def test_single_rank_runs_in_process():
    r, lines = _run(["--steps", "3", "--warmup", "1", "--ramp-ms", "1", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(lines[-1])
    assert j["n_gpus"] == 1 and "kernel_ms_per_rank" not in j["extra"]
    assert j["extra"]["steady_check"]["steps"] >= 3 and j["extra"]["cold_ms_per_step"] > 0


def test_launcher_does_not_import_torch_or_hip():
    """the parent of the ranks must stay clear of the GPU: it may not import torch or load the extension"""
    code = (
        "import sys, os; sys.argv=['bench.py','--gpus','2']; os.environ.pop('WORLD_SIZE', None)\n"
        "import bench, subprocess\n"
        "calls=[]\n"
        "class P:\n"
        "    stdout=[]\n"
        "    def wait(self): return 0\n"
        "subprocess.Popen=lambda cmd, **k: (calls.append(cmd), P())[1]\n"
        "subprocess.run=lambda cmd, **k: type('R',(),{'returncode':0,'stdout':'8\\n'})()\n"
        "try:\n"
        "    bench.main()\n"
        "except SystemExit as e:\n"
        "    assert e.code == 0, e.code\n"
        "assert 'torch' not in sys.modules and 'directdemod_amd' not in sys.modules, sorted(m for m in sys.modules if 'torch' in m)[:5]\n"
        "cmd=calls[0]\n"
        "assert '-m' in cmd and 'torch.distributed.run' in cmd and '--nproc-per-node=2' in cmd and '127.0.0.1' in cmd, cmd\n"
        "assert cmd[-2:] == ['--gpus','2'], cmd\n"
        "print('ok')\n")
    env = dict(os.environ)
    env.pop("DD_BENCH_STUB", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=60)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_the_drivers_8_gpu_command_line_parses_and_keeps_the_contract():
    """The scaling record is the driver's to measure (`python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8
    --steps K --warmup W`); what can be checked here is that this command line parses to the C2 shard per rank, that the
    launcher would pass every argument through unchanged, and that `extra.rccl_init_s` (rendezvous + first collective) is part
    of a multi-rank line."""
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert (args.gpus, args.steps, args.warmup, args.log2n) == (8, 20, 5, 26)
    assert args.steady_ms >= 6000 and not args.no_gather
    args = bench.parse_args(["--gpus", "8", "--log2n", "26", "--steps", "20", "--warmup", "5", "--no-gather"])
    assert args.log2n == 26 and args.no_gather
    with pytest.raises(SystemExit):
        bench.parse_args(["--gpus", "eight"])
    r, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--ramp-ms", "2"])
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-2000:]
    j = json.loads(lines[0])
    assert 0 <= j["extra"]["rccl_init_s"] < 60 and j["extra"]["world_size_seen"] == 2
    assert "[bench] rank 0/2: init_process_group" in r.stderr
