"""The N > 1 path on real hardware: RCCL process group through torch.distributed.run, as bench.py --gpus N starts it
(reference: DDP over all visible GPUs, main.py:64-70,81).  Two ranks when the box has two GPUs, otherwise one rank --
the collectives then still run through RCCL (a group of one), which is all a 1-GPU box can prove."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(n, script, *args, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script, *args]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(900)
def test_rccl_process_group_grad_allreduce_and_syncbn():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    n = 2 if torch.cuda.device_count() >= 2 else 1
    r = _run_ranks(n, os.path.join(ROOT, "tests", "rccl_worker.py"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert f"RCCL_OK world={n} backend=nccl" in r.stdout


@pytest.mark.timeout(1200)
def test_bench_gpus_flag_starts_that_many_ranks_or_fails():
    """`python bench.py --gpus 2` (no launcher): on a box with >= 2 GPUs it starts 2 RCCL ranks and prints n_gpus: 2; on a
    1-GPU box it exits non-zero instead of printing a 1-rank number under a 2-GPU label."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--workload", "cfg1_plumbing",
           "--no-cpu-baseline", "--no-strict-fp32", "--sustain", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1000, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["value"] > 0
    else:
        assert r.returncode != 0 and "only 1 GPU" in (r.stderr + r.stdout)
