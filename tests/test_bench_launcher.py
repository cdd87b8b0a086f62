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
    r, lines = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--ramp-ms", "2", "--gather"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 1
    assert j["data"] == "stub" and j["config"]["kernel_path"] == "stub"
    assert j["scaling"] == "weak" and j["higher_is_better"] is True
    assert len(j["extra"]["kernel_ms_per_rank"]) == 2
    # gather leg: rank 0 of the stream contributes n-1 outputs (quirk Q3), every other rank n
    assert j["extra"]["gathered_outputs"] == 2 * j["config"]["samples_per_gpu"] - 1
    assert "with_all_gather_MSamples_per_s" in j["extra"]
    assert "cpu_baseline" not in j                      # rank 0 at N = 1 only


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
        "subprocess.run=lambda cmd, **k: type('R',(),{'returncode':0})()\n"
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
