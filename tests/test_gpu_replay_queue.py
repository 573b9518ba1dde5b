"""-m gpu: replay rollouts on a device-resident queue of the sampler's speculated draws (DCD_REPLAY_QUEUE, round 4).

Reference: adversarial_runner.py:551-558 draws a replay level per finished env, in env order, between two steps
(level_sampler.py:664-680).  The pieces: `dcd_replay_pop` (queue entries to finished envs in env order), the masked level resets
from the device level table (`dcd_mg_reset_to_encoding_rows` / `_actions_rows`) and the runner's `_queued_replay_rollout` — the
whole run must equal the per-step host path bit for bit: decisions, every env's seeds, sampler state, np.random position, weights;
also when the queue runs dry and the rollout is rolled back and redone on the host path."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,K,p", [(37, 64, 0.3), (4096, 300, 0.05), (5000, 5000, 0.9), (1024, 0, 0.5)])
def test_replay_pop_hands_out_queue_entries_in_env_order(B, K, p):
    import dcd_hip
    rng = np.random.default_rng(B + K)
    info = (rng.random(B) < p).astype(np.uint8) * dcd_hip.INFO_EPISODE | (rng.integers(0, 2, B).astype(np.uint8) * dcd_hip.INFO_TRUNCATED)
    rows, seeds = rng.integers(0, 10**6, max(K, 1)).astype(np.int32), rng.integers(1, 2**31 - 1, max(K, 1)).astype(np.int32)
    start = 7 if K > 20 else 0
    t = lambda a: torch.from_numpy(a).to(DEV)
    d = dict(info=t(info), rows=t(rows), seeds=t(seeds), state=torch.tensor([start, 0], dtype=torch.int32, device=DEV),
             lvl=torch.full((B,), -7, dtype=torch.int32, device=DEV), cur=torch.arange(B, dtype=torch.int32, device=DEV),
             solved=torch.zeros(B, dtype=torch.uint8, device=DEV))
    dcd_hip.check(dcd_hip.lib().dcd_replay_pop(dcd_hip.ptr(d["info"]), C.c_int32(B), dcd_hip.ptr(d["rows"]), dcd_hip.ptr(d["seeds"]), C.c_int32(K),
                                               C.c_void_p(d["state"].data_ptr()), C.c_void_p(d["state"].data_ptr() + 4), dcd_hip.ptr(d["lvl"]),
                                               dcd_hip.ptr(d["cur"]), dcd_hip.ptr(d["solved"]), dcd_hip.stream_ptr()))
    fin = np.flatnonzero(info & dcd_hip.INFO_EPISODE)
    exp_lvl, exp_cur, exp_solved = np.full(B, -1, np.int32), np.arange(B, dtype=np.int32), np.zeros(B, np.uint8)
    for k, env in enumerate(fin):
        if start + k < K:
            exp_lvl[env], exp_cur[env], exp_solved[env] = rows[start + k], seeds[start + k], 1
    cursor, overflow = d["state"].cpu().tolist()
    assert cursor == start + len(fin) and overflow == int(start + len(fin) > K)
    assert np.array_equal(d["lvl"].cpu().numpy(), exp_lvl) and np.array_equal(d["cur"].cpu().numpy(), exp_cur)
    assert np.array_equal(d["solved"].cpu().numpy(), exp_solved)


@pytest.mark.parametrize("kind", ["encoding", "actions"])
def test_masked_table_reset_equals_the_indexed_reset(kind):
    """reset_to_table_rows(table, level_idx) == reset_to_encoding / reset_to_actions(table[rows], idx) on the chosen envs;
    every other env — record, metrics, outputs — is left alone."""
    import dcd_hip
    from tests import mg_diff
    B, R = 300, 40
    cfg = dict(n_clutter=25, choose_goal_last=1, engine_seed=3, replicas=2)
    eng = dcd_hip.MultiGridEngine(B, device=DEV, **cfg)
    ref = dcd_hip.MultiGridEngine(B, device=DEV, **cfg)
    for e in (eng, ref):
        e.reset_random()
    rng = np.random.default_rng(5)
    lvl = np.full(B, -1, np.int32)
    chosen = np.sort(rng.choice(B, 70, replace=False))
    lvl[chosen] = rng.integers(0, R, len(chosen))
    if kind == "encoding":
        table = torch.from_numpy(mg_diff.random_encodings(rng, R, 15, exotic=True)).to(DEV)
        lens = None
    else:
        table = torch.from_numpy(rng.integers(0, 169, (R, 27))).to(DEV)
        lens = torch.from_numpy(rng.integers(20, 28, R).astype(np.int32)).to(DEV)
    out = eng.student_buffers(B)
    for v in out.values():
        v.fill_(99)
    eng.reset_to_table_rows(table, lens, torch.from_numpy(lvl).to(DEV), out)
    idx = torch.from_numpy(chosen.astype(np.int32)).to(DEV)
    rows = torch.from_numpy(lvl[chosen].astype(np.int64)).to(DEV)
    if kind == "encoding":
        exp = ref.reset_to_encoding(table.index_select(0, rows), idx)
    else:
        exp = ref.reset_to_actions(table.index_select(0, rows), lens.index_select(0, rows), idx)
    assert torch.equal(eng.get_encodings(), ref.get_encodings()) and torch.equal(eng.get_state(), ref.get_state())
    for k in ("n_clutter_placed", "passable", "shortest_path_length", "distance_to_goal"):
        assert torch.equal(eng.get_metrics()[k], ref.get_metrics()[k]), k
    untouched = np.setdiff1d(np.arange(B), chosen)
    for k, v in out.items():
        assert torch.equal(v[idx.long()], exp[k]), k
        assert bool((v[torch.from_numpy(untouched).to(DEV)] == 99).all()), k
    assert eng.poll_errors() == ref.poll_errors()


def _run(monkeypatch, queue, stream, updates=10, **kw):
    from tests.test_gpu_runner import build, make_args
    monkeypatch.setenv("DCD_REPLAY_QUEUE", queue)
    monkeypatch.setenv("DCD_PREFETCH_STREAM", stream)
    a = dict(ued_algo="domain_randomization", use_plr=True, use_reset_random_dr=True, no_exploratory_grad_updates=True,
             level_replay_strategy="positive_value_loss", level_replay_score_transform="rank", level_replay_prob=0.7,
             level_replay_rho=0.2, level_replay_seed_buffer_size=100, staleness_coef=0.3, log_plr_buffer_stats=True,
             recurrent_adversary_env=False, num_steps=64, env_name="MultiGrid-MiniGoalLastAdversarial-v0")
    a.update(kw)
    runner, venv = build(make_args(**a))
    np.random.seed(4)
    log = []
    for j in range(updates):
        s = runner.run()
        st = runner.agents["agent"].storage
        log.append((bool(runner.sampled_level_info["level_replay"]), list(runner.current_level_seeds), s["agent_value_loss"],
                    s.get("agent_plr_passable_mass"), st.level_seeds.clone(), st.obs_u8["image"].clone() if st.obs_u8 is not None else None,
                    st.rewards.clone()))
    sampler = runner.level_samplers["agent"]
    out = dict(log=log, stale=sampler.seed_staleness.copy(), scores=sampler.seed_scores.copy(), rng=np.random.get_state()[1].copy(),
               w=torch.cat([p.detach().flatten() for p in runner.agents["agent"].algo.actor_critic.parameters()]).clone(),
               queued=getattr(runner, "replay_queue_rollouts", 0), overflows=getattr(runner, "replay_queue_overflows", 0))
    venv.close()
    return out


def _same(a, b):
    for x, y in zip(a["log"], b["log"]):
        assert x[0] == y[0] and x[1] == y[1] and x[2] == y[2]
        assert torch.equal(x[4], y[4]) and torch.equal(x[6], y[6]) and (x[5] is None or torch.equal(x[5], y[5]))
    assert np.array_equal(a["stale"], b["stale"]) and np.array_equal(a["scores"], b["scores"]) and np.array_equal(a["rng"], b["rng"])
    assert torch.equal(a["w"], b["w"])


@pytest.mark.parametrize("kw", [dict(), dict(use_reset_random_dr=False)])
def test_queued_replay_rollouts_equal_the_host_path(monkeypatch, kw):
    """Byte-encoded levels (use_reset_random_dr) and teacher-action levels (the random teacher's strings)."""
    host = _run(monkeypatch, "0", "3", **kw)
    dev = _run(monkeypatch, "1", "8", **kw)
    assert host["queued"] == 0 and dev["queued"] >= 3 and dev["overflows"] == 0     # >= 3: eager, captured, replayed from the graph
    assert sum(x[0] for x in host["log"]) >= 2
    _same(host, dev)


def test_a_dry_queue_rolls_back_and_redoes_the_rollout_on_the_host_path(monkeypatch):
    """One speculated draw per env is not enough for a 64-step rollout of 8-step episodes: the pop kernel flags the overflow,
    the batch is restored from its snapshot and the per-step path redoes the rollout — same run."""
    host = _run(monkeypatch, "0", "3")
    dry = _run(monkeypatch, "1", "1")
    assert dry["overflows"] >= 1
    _same(host, dry)
