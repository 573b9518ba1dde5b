"""-m gpu: DCD_CONCURRENT_PPO=1 — the three PPO updates of a PAIRED run() issued together from three host threads on three
streams.  Opt-in since round 3 and only under TENSILE_STREAMK_DATA_PARALLEL=1 (read by hipBLASLt once per process): with the
library's default StreamK GEMMs concurrent streams deadlock on the GPU (profiles/r03_concurrent_ppo_hang.log).
The checks therefore run in a CHILD process started with that variable (`python tests/test_gpu_concurrent_ppo.py`)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def check_concurrent_equals_in_order():
    """One PPO update is run-to-run bit-deterministic on this path and the minibatch permutations are drawn in program order
    before the threads start, so concurrent and in-order execution must agree EXACTLY: same seeds -> same losses, same
    gradient norms, same weights after two updates."""
    from tests.test_gpu_runner import build, make_args
    out = {}
    for mode in ("0", "1"):
        os.environ["DCD_CONCURRENT_PPO"] = mode
        args = make_args(ued_algo="paired", ppo_epoch=2, adv_ppo_epoch=2, num_mini_batch=2)
        runner, venv = build(args)
        assert runner._concurrent_updates == (mode == "1")
        stats = [runner.run() for _ in range(2)]
        assert not runner._pending_updates
        out[mode] = (stats, {n: {k: v.clone() for k, v in a.algo.actor_critic.state_dict().items()} for n, a in runner.agents.items()},
                     runner.agents["agent"].storage.actions.clone())
        venv.close()
    (s0, w0, a0), (s1, w1, a1) = out["0"], out["1"]
    assert torch.equal(a0, a1)
    for j in range(2):
        for k in ("mean_env_return", "total_episodes", "agent_value_loss", "adversary_value_loss", "adversary_env_value_loss",
                  "agent_pg_loss", "adversary_env_pg_loss", "agent_grad_norm", "adversary_grad_norm", "adversary_env_grad_norm"):
            assert np.isfinite(s1[j][k]) and s1[j][k] == s0[j][k], (j, k, s0[j][k], s1[j][k])
    for n in w0:
        for k in w0[n]:
            assert torch.equal(w1[n][k], w0[n][k]), f"{n}.{k}"


def check_same_seed_reproduces():
    """Two concurrent runs from the same --seed: identical stats and weights (the permutations used to be drawn inside the
    worker threads from the shared global generator)."""
    from tests.test_gpu_runner import build, make_args
    os.environ["DCD_CONCURRENT_PPO"] = "1"
    res = []
    for _ in range(2):
        args = make_args(ued_algo="paired", ppo_epoch=3, adv_ppo_epoch=3, num_mini_batch=4)
        runner, venv = build(args)
        stats = [runner.run() for _ in range(2)]
        res.append((stats[-1], {n: {k: v.clone() for k, v in a.algo.actor_critic.state_dict().items()} for n, a in runner.agents.items()}))
        venv.close()
    for k in ("agent_value_loss", "adversary_value_loss", "adversary_env_value_loss", "mean_env_return"):
        assert res[0][0][k] == res[1][0][k], k
    for n in res[0][1]:
        for k in res[0][1][n]:
            assert torch.equal(res[0][1][n][k], res[1][1][n][k]), f"{n}.{k}"


def check_the_shape_that_deadlocked():
    """PAIRED at 4096 envs x 8 steps: with StreamK GEMMs the first concurrent update never returned."""
    from tests.test_gpu_runner import build, make_args
    os.environ["DCD_CONCURRENT_PPO"] = "1"
    runner, venv = build(make_args(ued_algo="paired", num_processes=4096, num_steps=8))
    for _ in range(6):
        stats = runner.run()
    assert np.isfinite(stats["agent_value_loss"]) and np.isfinite(stats["adversary_env_value_loss"])
    venv.close()


def test_concurrent_updates_in_a_process_with_data_parallel_gemms():
    env = dict(os.environ, TENSILE_STREAMK_DATA_PARALLEL="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    assert r.returncode == 0 and "concurrent-ppo checks ok" in r.stdout, r.stdout[-3000:]


def test_concurrent_updates_are_refused_with_streamk_gemms(monkeypatch):
    from tests.test_gpu_runner import build, make_args
    monkeypatch.setenv("DCD_CONCURRENT_PPO", "1")
    monkeypatch.delenv("TENSILE_STREAMK_DATA_PARALLEL", raising=False)
    with pytest.raises(ValueError, match="TENSILE_STREAMK_DATA_PARALLEL"):
        build(make_args(ued_algo="paired"))
    # set from Python AFTER the process started: the library has read its value already, the guard looks at the start-up block
    monkeypatch.setenv("TENSILE_STREAMK_DATA_PARALLEL", "1")
    with pytest.raises(ValueError, match="TENSILE_STREAMK_DATA_PARALLEL"):
        build(make_args(ued_algo="paired"))
    monkeypatch.delenv("DCD_CONCURRENT_PPO")
    runner, venv = build(make_args(ued_algo="paired"))
    assert not runner._concurrent_updates                           # the default: program order
    venv.close()


if __name__ == "__main__":
    for p in (ROOT, os.path.join(ROOT, "dcd-isaac_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    assert os.environ.get("TENSILE_STREAMK_DATA_PARALLEL") == "1"
    check_concurrent_equals_in_order()
    check_same_seed_reproduces()
    check_the_shape_that_deadlocked()
    print("concurrent-ppo checks ok")
