"""-m gpu: whole rollouts captured in ONE HIP graph each (round 1 chunked them to 32 steps after a GPU abort whose cause is
now known: torch.multinomial's validity check rests on PyTorch's multi-block full reductions, which return stale partial
results when replayed from a HIP graph on this stack — DESIGN.md §8, tools/graph_node_order_probe.py)."""
import numpy as np
import pytest
import torch

from tests.test_gpu_runner import build, make_args

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode,fused", [("2", "0"), ("1", "1")])
def test_full_length_rollouts_replay_from_single_graphs(monkeypatch, mode, fused):
    """256-step PAIRED rollouts, policy as PyTorch modules (DCD_HIP_GRAPHS=2: teacher and students captured, ~12k kernel
    nodes per student graph) and on the fused HIP path: one graph per rollout, four updates (eager, capture + replay, two
    replays); under capture no draw may go through torch.multinomial; everything the policies wrote stays finite and
    normalised, actions stay in range, weights stay finite."""
    monkeypatch.setenv("DCD_HIP_GRAPHS", mode)
    monkeypatch.setenv("DCD_FUSED_ACT", fused)
    monkeypatch.setenv("DCD_INCREMENTAL_ACT", "0")
    real = torch.multinomial

    def guarded(*a, **kw):
        assert not torch.cuda.is_current_stream_capturing(), "torch.multinomial inside a HIP-graph capture"
        return real(*a, **kw)
    monkeypatch.setattr(torch, "multinomial", guarded)
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
    venv.close()
