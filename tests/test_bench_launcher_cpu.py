"""CPU: `python bench.py --gpus N` with no launcher starts N ranks itself (bench.py::self_launch) — the parent makes no GPU call,
sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* for each child, relays rank 0's line and exits with the worst child's code.
Without a HIP device every rank refuses to run (the product has no CPU mode): rc 1, nothing on stdout, one refusal per rank."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None):
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    base.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=base, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=300)


def test_parent_starts_one_child_per_rank_and_reports_their_failure():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the GPU variant is tests/test_gpu_bench_contract.py")
    r = _run("--gpus", "3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-train-leg", "--no-sweep")
    assert r.returncode == 1 and r.stdout.strip() == ""
    assert r.stderr.count("needs a HIP device") == 3                # every rank got as far as its own device check
    for k in range(3):
        assert f"rank {k} exited with code 1" in r.stderr
    assert "torch.distributed.run" not in r.stderr                  # round 3's refusal is gone


def test_mismatched_launcher_world_size_is_refused():
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in r.stderr
