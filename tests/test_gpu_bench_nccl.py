"""
The RCCL leg of bench.py on the 1-GPU test box: one rank, but through the distributed branch (DD_BENCH_FORCE_DIST=1) --
torch.distributed.init_process_group("nccl") (nccl IS RCCL on ROCm), the barriers, the variable-count gather of the decoded
stream and the timed fixed-size all_gather -- so that the code path the driver's 8-GPU run takes has executed at least once on
real hardware before that node is the first place it runs.  No scaling number comes out of this.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(150)
def test_one_rank_through_the_nccl_branch():
    env = dict(os.environ)
    for k in ("DD_BENCH_STUB", "DD_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                "DD_BENCH_FORCE_DIST": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--log2n", "22", "--steps", "3",
                        "--warmup", "1", "--ramp-ms", "5", "--steady-ms", "50", "--no-cpu-baseline", "--no-side"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=120)
    sys.stderr.write("".join(ln + "\n" for ln in r.stderr.splitlines() if ln.startswith("[bench]")))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    n = 1 << 22
    assert j["n_gpus"] == 1 and j["config"]["samples_per_gpu"] == n
    assert j["extra"]["backend"] == "nccl" and j["extra"]["world_size_seen"] == 1
    assert 0 <= j["extra"]["rccl_init_s"] < 60                     # rendezvous + communicator set-up, as the 8-GPU record will show it
    assert j["extra"]["gathered_outputs"] == n - 1                 # rank 0 of a stream owns one output fewer (quirk Q3)
    assert j["extra"]["with_all_gather_MSamples_per_s"] > 0 and j["extra"]["all_gather_ms_per_step"] > 0
    assert j["config"]["kernel"] == "k_chain_cos1k" and j["data"] == "synthetic"
    assert 0.005 < j["extra"]["output_rms_rad"] < 0.02             # the demodulated 1 kHz tone (deviation 5 rad)
