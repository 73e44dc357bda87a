"""The launcher path of bench.py on the CPU: `bench.py --gpus N` starts its own ranks.

With --rehearse-cpu the ranks do everything around the kernels (rendezvous, barriers, the
max-over-ranks reduction, the chunked gather of result rows to rank 0) over gloo and exchange
synthetic rows; stdout must carry exactly one JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("world", [2, 3])
def test_bench_starts_its_own_ranks_and_gathers(world):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(world), "--rehearse-cpu"], capture_output=True, text=True,
                       timeout=240, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["rehearsal"] is True and rec["n_gpus"] == world and rec["gather_ok"] is True


def test_bench_fails_when_gpus_are_missing():
    """Without GPUs the launcher stops at its device check, before starting any rank."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this host has the GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--no-cpu"], capture_output=True, text=True, timeout=240,
                       env=_env())
    assert r.returncode == 2 and "needs 2 GPUs" in r.stderr and r.stdout.strip() == ""


def test_bench_rank_failure_ends_the_run():
    """A rank that dies takes the launch down with a non-zero exit instead of leaving the others waiting."""
    env = _env()
    env["SLAM_AMD_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse-cpu"], capture_output=True, text=True,
                       timeout=240, env=env)
    assert r.returncode != 0
    assert "rank 1 exited" in r.stderr
