"""-m gpu: whole rollouts captured in ONE HIP graph each (round 1 chunked them to 32 steps after a GPU abort whose cause is
now known: torch.multinomial's validity check rests on PyTorch's multi-block full reductions, which return stale partial
results when replayed from a HIP graph on this stack — DESIGN.md §8, tools/graph_node_order_probe.py)."""
import numpy as np
import pytest
import torch

from tests.test_gpu_runner import build, make_args

pytestmark = pytest.mark.gpu


def guard_captured_reductions(monkeypatch):
    """Under capture: no torch.multinomial and no dim-less (full) reduction — PyTorch's multi-block full reductions return stale
    partials when replayed from a HIP graph on this stack, silently: a captured one must fail the test, not
    corrupt a rollout.  Returns the list the violations are recorded in."""
    seen = []
    real = torch.multinomial

    def guarded(*a, **kw):
        assert not torch.cuda.is_current_stream_capturing(), "torch.multinomial inside a HIP-graph capture"
        return real(*a, **kw)
    monkeypatch.setattr(torch, "multinomial", guarded)
    for name in ("max", "min", "sum", "mean", "any", "all", "prod", "std", "var", "norm"):
        orig = getattr(torch.Tensor, name)

        def method(self, *a, _orig=orig, _name=name, **kw):
            full = not a and kw.get("dim") is None and not (_name == "norm" and kw.get("p") is not None and kw.get("dim") is not None)
            if full and self.is_cuda and self.numel() > 1 and torch.cuda.is_current_stream_capturing():
                seen.append((_name, tuple(self.shape)))
            return _orig(self, *a, **kw)
        monkeypatch.setattr(torch.Tensor, name, method)
    return seen


@pytest.mark.parametrize("mode,fused", [("2", "0"), ("1", "1")])
def test_full_length_rollouts_replay_from_single_graphs(monkeypatch, mode, fused):
    """256-step PAIRED rollouts, policy as PyTorch modules (DCD_HIP_GRAPHS=2: teacher and students captured, ~12k kernel
    nodes per student graph) and on the fused HIP path: one graph per rollout, four updates (eager, capture + replay, two
    replays); under capture no draw may go through torch.multinomial; everything the policies wrote stays finite and
    normalised, actions stay in range, weights stay finite."""
    monkeypatch.setenv("DCD_HIP_GRAPHS", mode)
    monkeypatch.setenv("DCD_FUSED_ACT", fused)
    monkeypatch.setenv("DCD_INCREMENTAL_ACT", "0")
    seen = guard_captured_reductions(monkeypatch)
    args = make_args(ued_algo="paired", num_processes=256, num_steps=256)
    runner, venv = build(args)
    for it in range(4):
        stats = runner.run()
        for name, n_act in (("agent", 7), ("adversary_agent", 7), ("adversary_env", 169)):
            st = runner.agents[name].storage
            ld = st.action_log_dist
            assert torch.isfinite(ld).all(), (it, name)
            assert torch.allclose(ld.exp().sum(-1), torch.ones_like(ld[..., 0]), atol=1e-4), (it, name)
            assert int(st.actions.min()) >= 0 and int(st.actions.max()) < n_act
            assert torch.isfinite(st.value_preds).all() and torch.isfinite(st.recurrent_hidden_states).all()
            assert all(bool(torch.isfinite(p).all()) for p in runner.agents[name].algo.actor_critic.parameters())
        assert np.isfinite(stats["agent_value_loss"]) and np.isfinite(stats["adversary_env_value_loss"])
    assert runner._graphs and all(isinstance(g, list) and len(g) == 1 for g in runner._graphs.values())
    if mode == "2":
        assert any(k[0] == "teacher" for k in runner._graphs)
    assert not seen, f"full reductions captured into a HIP graph: {seen[:5]}"
    venv.close()


def test_module_rollouts_at_the_batch_that_aborted(monkeypatch):
    """Round 1's abort happened at 4096 envs: the teacher's 169-way softmax is a 692k-element tensor there, whose full
    reductions run as multi-block kernels.  The PyTorch-module policy path (DCD_HIP_GRAPHS=2, dense teacher operand) at that
    batch, short student rollouts: capture + three replays, no multinomial and no full reduction under capture, every draw
    in range and every stored distribution normalised."""
    monkeypatch.setenv("DCD_HIP_GRAPHS", "2")
    monkeypatch.setenv("DCD_FUSED_ACT", "0")
    monkeypatch.setenv("DCD_INCREMENTAL_ACT", "0")
    seen = guard_captured_reductions(monkeypatch)
    args = make_args(ued_algo="paired", num_processes=4096, num_steps=8)
    runner, venv = build(args)
    for it in range(5):
        runner.run()
        for name, n_act in (("agent", 7), ("adversary_agent", 7), ("adversary_env", 169)):
            st = runner.agents[name].storage
            ld = st.action_log_dist
            assert torch.isfinite(ld).all(), (it, name)
            assert torch.allclose(ld.exp().sum(-1), torch.ones_like(ld[..., 0]), atol=1e-4), (it, name)
            assert int(st.actions.min()) >= 0 and int(st.actions.max()) < n_act
    assert any(k[0] == "teacher" for k in runner._graphs)
    assert not seen, f"full reductions captured into a HIP graph: {seen[:5]}"
    venv.close()
