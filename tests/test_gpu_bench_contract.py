"""-m gpu: bench.py's contract with the driver — ONE JSON line on stdout, whatever the libraries print (RCCL writes a five-line
version banner to stdout when the first communicator is created), carrying the metric, the roofline object with live per-kernel
event times, and — here — the collectives of the multi-rank line executed through a 1-rank RCCL group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_json_line_with_roofline_and_kernel_times():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-train-leg",
                        "--no-sweep", "--rccl-single"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "kernel_times"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "env-steps/s" and d["value"] > 1e8 and "workload" in d["config"]
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    assert abs(ro["achieved"] - ro["algorithmic_bytes_per_launch"] / (ro["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * ro["achieved"]
    kt = d["kernel_times"]["kernels"]
    assert kt["mg_step_kernel"]["launches_per_pass"] == 256 and kt["mg_level_kernel"]["launches_per_pass"] == 30
    assert 0.9 < d["kernel_times"]["accounts_for"] < 1.15                 # the per-kernel event times account for the replayed pass
    for name in ("mg_step_kernel", "mg_level_kernel", "rollout_tail_kernel", "seg_score_lds_kernel"):   # HBM GB/s per kernel
        k = kt[name]
        assert abs(k["achieved_GBs"] - k["algorithmic_bytes_per_launch"] / (k["median_us"] * 1e-6) / 1e9) < 1e-6 * k["achieved_GBs"]
        assert 0 < k["frac_of_hbm_peak"] < 1
    assert d["grad_exchange"]["bytes"] > 9e7                               # the gradient all-reduce leg ran through RCCL


def test_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` as the driver records it (no torch.distributed.run): the parent starts the two ranks itself —
    here both on the box's one GPU over gloo — and relays rank 0's single line (VERDICT r3 missing #1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-train-leg", "--no-sweep"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    # (a rehearsal of the launch path: two ranks share the one GPU and exchange over gloo through the host — 1.8e7 here, no figure of merit)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 1e6
