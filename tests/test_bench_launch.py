"""`python bench.py --gpus N` must work as typed: without WORLD_SIZE it starts the N
ranks itself, before any GPU call.  Checked here without a GPU: the ranks rendezvous over gloo on
127.0.0.1 and rank 0 prints the one JSON line; and under torch.distributed.run the same file is
one of the ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "HSA_ENABLE_IPC_MODE_LEGACY")}
    return env


@pytest.mark.parametrize("n", [2, 4])
def test_self_launch_starts_n_ranks_and_prints_one_line(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--launch-check"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["rank_sum"] == n * (n - 1) / 2 == out["local_rank_sum"]
    assert out["hsa_enable_ipc_mode_legacy"] == "0"  # the same RCCL environment in both launch forms


def test_torchrun_form_still_works():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", BENCH, "--gpus", "2",
                        "--launch-check"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    assert json.loads(lines[0])["hsa_enable_ipc_mode_legacy"] == "0"


def test_a_failing_rank_ends_the_launch():
    """--gpus 2 on a machine without two GPUs: the ranks refuse, the launcher returns non-zero
    instead of hanging."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are visible")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr
