"""-m gpu: AdversarialRunner end to end on small batches (PAIRED, DR, robust PLR, ACCEL, minimax), and
the fused device-level rollout against the reference's host-side step loop restated on top of the
reference-compatible `step_env` API (adversarial_runner.py:497-588)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_args(**kw):
    from arguments import parser
    args = parser.parse_args([])
    base = dict(env_name="MultiGrid-GoalLastFewerBlocksAdversarial-v0", num_processes=64, num_steps=48, ppo_epoch=1,
                adv_ppo_epoch=1, handle_timelimits=True, recurrent_adversary_env=True, log_grad_norm=True, seed=3)
    base.update(kw)
    for k, v in base.items():
        assert hasattr(args, k), k
        setattr(args, k, v)
    return args


def build(args):
    from envs.runners import AdversarialRunner
    from util import create_parallel_env, make_agent, make_plr_args, seed
    seed(args.seed)
    venv, ued_venv = create_parallel_env(args, device=DEV)
    agent = make_agent("agent", venv, args, DEV)
    adversary_agent = make_agent("adversary_agent", venv, args, DEV) if args.ued_algo in ("paired", "flexible_paired") else None
    adversary_env = make_agent("adversary_env", venv, args, DEV)
    if args.ued_algo == "domain_randomization" and args.use_plr and not args.use_reset_random_dr:
        adversary_env.random()                                         # train.py:84-86
    plr_args = make_plr_args(args, venv.observation_space, venv.action_space) if args.use_plr else None
    runner = AdversarialRunner(args=args, venv=venv, agent=agent, ued_venv=ued_venv, adversary_agent=adversary_agent,
                               adversary_env=adversary_env, train=True, plr_args=plr_args, device=DEV)
    return runner, venv


REF_KEYS = ["steps", "total_episodes", "total_seeds", "total_student_grad_updates", "mean_agent_return", "agent_value_loss",
            "agent_pg_loss", "agent_dist_entropy", "mean_adversary_agent_return", "adversary_value_loss",
            "adversary_pg_loss", "adversary_dist_entropy"]


@pytest.mark.parametrize("algo", ["paired", "flexible_paired", "minimax"])
def test_paired_family_runs(algo):
    args = make_args(ued_algo=algo)
    runner, venv = build(args)
    before = {k: v.clone() for k, v in runner.agents["adversary_env"].algo.actor_critic.state_dict().items()}
    for j in range(2):
        stats = runner.run()
    for k in REF_KEYS + ["mean_env_return", "adversary_env_pg_loss", "num_blocks", "passable_ratio", "shortest_path_length"]:
        assert k in stats, k
    assert stats["steps"] == 2 * 64 * 48 and stats["total_seeds"] == 2 * 64
    assert np.isfinite(stats["agent_value_loss"]) and np.isfinite(stats["adversary_env_value_loss"])
    assert 0 <= stats["mean_env_return"] <= 1 or algo == "minimax"
    after = runner.agents["adversary_env"].algo.actor_critic.state_dict()
    assert any(not torch.equal(before[k], after[k]) for k in before)          # teacher learned from the regret
    sd = runner.state_dict()
    assert set(sd) >= {"agent_state_dict", "optimizer_state_dict", "num_updates", "level_store", "level_samplers"}
    assert sd["num_updates"] == 2
    venv.close()


def test_domain_randomization_runs():
    args = make_args(ued_algo="domain_randomization", recurrent_adversary_env=False)
    runner, venv = build(args)
    s = [runner.run() for _ in range(2)][-1]
    assert s["total_seeds"] >= 2 * 64 and s["total_episodes"] > 0 and np.isfinite(s["agent_value_loss"])
    venv.close()


def test_robust_plr_runs_and_fills_buffer():
    args = make_args(ued_algo="domain_randomization", use_plr=True, use_reset_random_dr=True, no_exploratory_grad_updates=True,
                     level_replay_strategy="grounded_signed_value_loss", level_replay_score_transform="rank",
                     level_replay_prob=0.5, level_replay_rho=0.5, level_replay_seed_buffer_size=100, staleness_coef=0.3,
                     log_plr_buffer_stats=True, recurrent_adversary_env=False, num_steps=64,
                     env_name="MultiGrid-MiniGoalLastAdversarial-v0")        # 6x6, 50-step episodes: levels complete
    runner, venv = build(args)
    np.random.seed(0)
    replays = 0
    for j in range(8):
        s = runner.run()
        replays += int(runner.sampled_level_info["level_replay"])
    sampler = runner.level_samplers["agent"]
    assert sampler.working_seed_buffer_size > 50 and replays >= 1
    assert len(runner.level_store) >= sampler.working_seed_buffer_size
    assert s["total_student_grad_updates"] == replays          # robust PLR: gradients only on replay levels
    assert "agent_plr_max_score" in s and np.isfinite(s["agent_plr_max_score"])
    w = sampler.sample_weights()
    assert abs(w.sum() - 1) < 1e-9 and (w >= 0).all()
    venv.close()


def test_prefetched_replay_draws_equal_the_sequential_order(monkeypatch):
    """DCD_PREFETCH_DRAWS (default on): the next update's replay decision and first draws are computed on a host thread under
    the PPO kernels of the current one (LevelSampler.prefetch_next_update) — every decision, every env's seeds, the sampler's
    state and the students' weights must equal the run that draws in program order (=0)."""
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DCD_PREFETCH_DRAWS", mode)
        args = make_args(ued_algo="domain_randomization", use_plr=True, use_reset_random_dr=True, no_exploratory_grad_updates=True,
                         level_replay_strategy="positive_value_loss", level_replay_score_transform="rank", level_replay_prob=0.7,
                         level_replay_rho=0.2, level_replay_seed_buffer_size=100, staleness_coef=0.3, log_plr_buffer_stats=True,
                         recurrent_adversary_env=False, num_steps=64, env_name="MultiGrid-MiniGoalLastAdversarial-v0")
        runner, venv = build(args)
        np.random.seed(4)
        log = []
        for j in range(10):
            s = runner.run()
            log.append((bool(runner.sampled_level_info["level_replay"]), list(runner.current_level_seeds), s["agent_plr_passable_mass"],
                        s["agent_value_loss"]))
        sampler = runner.level_samplers["agent"]
        out[mode] = (log, sampler.seed_staleness.copy(), sampler.seed_scores.copy(), np.random.get_state()[1].copy(),
                     torch.cat([p.detach().flatten() for p in runner.agents["agent"].algo.actor_critic.parameters()]).clone(),
                     getattr(sampler, "prefetch_hits", 0))
        venv.close()
    a, b = out["0"], out["1"]
    assert a[5] == 0 and b[5] >= 8                                  # speculation off / consumed
    assert sum(x[0] for x in a[0]) >= 2
    for x, y in zip(a[0], b[0]):
        assert x[0] == y[0] and x[1] == y[1] and x[3] == y[3] and (x[2] == y[2] or (x[2] != x[2] and y[2] != y[2]))
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and torch.equal(a[4], b[4])


def test_exploratory_updates_without_backward_equal_the_ones_with():
    """Robust PLR's exploratory updates discard their gradients; without gradient-norm logging the backward pass is skipped
    (PPO.update, forward-only).  Losses, decisions, seeds and weights over ten updates equal the run that computes them."""
    out = {}
    for log_norm in (True, False):
        args = make_args(ued_algo="domain_randomization", use_plr=True, use_reset_random_dr=True, no_exploratory_grad_updates=True,
                         level_replay_strategy="positive_value_loss", level_replay_score_transform="rank", level_replay_prob=0.5,
                         level_replay_rho=0.2, level_replay_seed_buffer_size=100, staleness_coef=0.3, recurrent_adversary_env=False,
                         num_steps=64, env_name="MultiGrid-MiniGoalLastAdversarial-v0", log_grad_norm=log_norm)
        runner, venv = build(args)
        np.random.seed(6)
        log = []
        for j in range(10):
            s = runner.run()
            log.append((bool(runner.sampled_level_info["level_replay"]), list(runner.current_level_seeds), s["agent_value_loss"],
                        s["agent_pg_loss"], s["agent_dist_entropy"]))
        out[log_norm] = (log, torch.cat([p.detach().flatten() for p in runner.agents["agent"].algo.actor_critic.parameters()]).clone())
        venv.close()
    assert out[True][0] == out[False][0] and torch.equal(out[True][1], out[False][1])
    assert any(x[0] for x in out[True][0]) and not all(x[0] for x in out[True][0])


def test_accel_edits_levels():
    args = make_args(ued_algo="domain_randomization", use_plr=True, use_reset_random_dr=True, no_exploratory_grad_updates=True,
                     use_editor=True, level_editor_prob=1.0, num_edits=5, base_levels="easy",
                     level_replay_strategy="positive_value_loss", level_replay_prob=0.8, level_replay_rho=0.3,
                     level_replay_seed_buffer_size=64, recurrent_adversary_env=False, num_steps=64,
                     env_name="MultiGrid-MiniGoalLastAdversarial-v0")
    runner, venv = build(args)
    np.random.seed(1)
    for j in range(8):
        s = runner.run()
    assert runner.total_num_edits >= 1
    assert max(len(v) for v in runner.level_store.seed2parent.values()) >= 1      # edited levels carry lineage
    venv.close()


def test_device_rollout_equals_reference_host_loop():
    """Same levels, same action tape: storage filled by the fused device path == storage filled by the
    reference's step loop (restated here) over the compat API."""
    from algos import RolloutStorage
    from envs.wrappers import HipAdversarialVecEnv
    from envs.registration import make
    import dcd_hip
    B, T = 96, 300
    mk = lambda: make("MultiGrid-GoalLastFewerBlocksAdversarial-v0", seed=0)
    va = HipAdversarialVecEnv([mk] * B, device=DEV, engine_seed=5)
    vb = HipAdversarialVecEnv([mk] * B, device=DEV, engine_seed=5)
    for v in (va, vb):
        v.set_seed(list(range(B)))
        v.reset()
        for k in range(27):
            loc = dcd_hip.engine.random_actions(B, 0, 2, k, 169, DEV)
            v.step_adversary_device(loc, {})
    sa = RolloutStorage(None, T, B, va.observation_space, va.action_space, 4, "lstm", use_proper_time_limits=True, device=DEV)
    sb = RolloutStorage(None, T, B, vb.observation_space, vb.action_space, 4, "lstm", use_proper_time_limits=True, device=DEV)
    acts = dcd_hip.engine.random_actions(B, 0, 3, 0, 7, DEV, T=T)
    # --- device path
    va.reset_agent_device(sa.student_obs_out(0))
    for t in range(T):
        va.step_env_device(acts[t], sa.step_out(t), last_step=(t == T - 1))
        sa.advance()
    # --- reference host loop (adversarial_runner.py:484-588) over the compat API
    obs = vb.reset_agent()
    sb.copy_obs_to_index(obs, 0)
    eps = 0
    for t in range(T):
        obs, reward, done, infos = vb.step_env(acts[t].cpu().view(-1, 1), reset_random=False)
        if t >= T - 1:
            for i, d in enumerate(done):
                if not d:
                    infos[i]["cliffhanger"] = True
                    infos[i]["truncated"] = True
                    infos[i]["truncated_obs"] = {k: obs[k][i] for k in obs}
            done = np.ones_like(done, dtype=float)
        for i, info in enumerate(infos):
            if "episode" in info:
                eps += 1
                if "truncated_obs" in info:
                    sb.insert_truncated_obs(info["truncated_obs"], index=i)
        masks = torch.FloatTensor([[0.0] if d else [1.0] for d in done])
        bad = torch.FloatTensor([[0.0] if "truncated" in info else [1.0] for info in infos])
        cliff = torch.FloatTensor([[0.0] if "cliffhanger" in info else [1.0] for info in infos])
        z = torch.zeros(B, 1)
        sb.insert(obs, (z.repeat(1, 4), z.repeat(1, 4)), z.long(), z, torch.zeros(B, 7), z, reward, masks, bad,
                  cliffhanger_masks=cliff)
    assert eps > 20
    for name in ("rewards", "masks", "bad_masks", "cliffhanger_masks"):
        assert torch.equal(getattr(sa, name), getattr(sb, name)), name
    for k in ("image", "direction"):
        assert torch.equal(sa.obs[k], sb.obs[k]), k
        assert torch.equal(sa.truncated_obs[k], sb.truncated_obs[k]), "truncated_" + k
    assert int((sa.ep_info & 1).sum()) == eps
    assert torch.equal(va.get_encodings_device(), vb.get_encodings_device())
    va.close(); vb.close()


def test_opaque_walls_env_trains():
    """MultiGrid-GoalLastFewerBlocksOpaqueWallsAdversarial-v0 (adversarial.py:655-658): views masked by
    Grid.process_vis — unseen cells are all-zero triples, which never occur with see-through walls."""
    args = make_args(ued_algo="paired", env_name="MultiGrid-GoalLastFewerBlocksOpaqueWallsAdversarial-v0")
    runner, venv = build(args)
    stats = runner.run()
    img = runner.agents["agent"].storage.obs["image"]                  # [T+1,B,3,5,5] = encoding / 10
    unseen = (img == 0).all(dim=2)
    assert unseen.any() and not unseen.all()
    assert not unseen[..., 2, 4].any()                               # the agent's own cell is always visible
    assert np.isfinite(stats["agent_value_loss"])
    venv.close()


def _shipped_configs():
    import json, os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mg_train_configs.json")
    return sorted(json.load(open(path)).items())


@pytest.mark.parametrize("name,flags", _shipped_configs())
def test_every_shipped_multigrid_config_trains(name, flags):
    """Drop-in check over the reference's own MultiGrid training configurations (train_scripts/grid_configs/
    minigrid/*/*.json -> `python -m train --k=v ...`, make_cmd.py:30-45): every flag parses with our
    arguments.parser and three updates of the configured algorithm (DR, minimax, PAIRED, PLR, REPAIRED,
    robust PLR, ACCEL) run on the GPU path.  Shrunk only in scale: 48-step rollouts, a 64-level replay buffer."""
    from arguments import parser
    argv = [f"--{k}={v}" for k, v in flags.items()]
    args = parser.parse_args(argv)
    for k, v in flags.items():                                      # what was parsed is what the config says
        got = getattr(args, k)
        assert (got == v) or (isinstance(v, float) and abs(got - v) < 1e-12) or (k == "log_dir"), (k, got, v)
    args.num_steps, args.seed = 48, 5
    if args.use_plr:
        args.level_replay_seed_buffer_size = 64
    runner, venv = build(args)
    np.random.seed(1)
    stats = [runner.run() for _ in range(3)][-1]
    assert stats["steps"] == (3 + runner.total_num_edits) * args.num_processes * 48
    assert np.isfinite(stats["agent_value_loss"]) and stats["total_episodes"] > 0
    if args.use_plr and args.level_replay_prob > 0:
        assert len(runner.level_store) > 0
    venv.close()


def test_graph_replayed_rollouts_equal_eager_rollouts(monkeypatch):
    """Student rollouts on the fused policy path are captured as HIP graphs and replayed (DCD_HIP_GRAPHS=1, default).
    Same seeds, graphs off vs on: every storage tensor bit-identical after each of four consecutive rollouts (the
    graph path runs eager once, captures on the second call, then only replays; the Philox counter base lives on the
    device so that replays draw fresh numbers), also across an optimiser step in between."""
    results = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DCD_HIP_GRAPHS", mode)
        args = make_args(ued_algo="paired", num_steps=40)
        runner, venv = build(args)
        assert runner.use_hip_graphs == (mode == "1")
        agent, teacher = runner.agents["agent"], runner.agents["adversary_env"]
        runner._teacher_rollout(teacher, runner.adversary_env_rollout_steps)
        snaps = []
        for k in range(4):
            runner._student_rollout(agent, 40, False, None, {})
            st = agent.storage
            snaps.append({n: getattr(st, n).clone() for n in ("actions", "action_log_probs", "action_log_dist", "value_preds",
                                                              "rewards", "masks", "bad_masks", "recurrent_hidden_states", "ep_ret")}
                         | {"image": st.obs["image"].clone(), "direction": st.obs["direction"].clone()})
            if k == 1:                                               # the policy changes between rollouts, as in training
                with torch.no_grad():
                    for p in agent.algo.actor_critic.parameters():
                        p.add_(0.01 * torch.sin(torch.arange(p.numel(), device=p.device, dtype=p.dtype)).view_as(p))
        results[mode] = snaps
        assert runner._fused_actor(agent) is not None
        venv.close()
    for k, (a, b) in enumerate(zip(results["0"], results["1"])):
        for n in a:
            assert torch.equal(a[n], b[n]), (k, n)
    assert not torch.equal(results["1"][0]["actions"], results["1"][1]["actions"])     # fresh draws per rollout
    assert not torch.equal(results["1"][2]["actions"], results["1"][3]["actions"])


def _storage_tensors(st):
    out = {k: v for k, v in vars(st).items() if isinstance(v, torch.Tensor)}
    for name in ("obs", "truncated_obs"):
        d = getattr(st, name)
        if isinstance(d, dict):
            out.update({f"{name}.{k}": v for k, v in d.items()})
    return out


@pytest.mark.parametrize("fused", ["1", "0"])
def test_paired_lockstep_rollouts_equal_the_two_sequential_rollouts(monkeypatch, fused):
    """DCD_PAIR_ROLLOUT: protagonist and antagonist advance in ONE env-step launch per step on two copies of the teacher's
    levels.  Same seeds, first update, policy on the fused HIP path (per-agent Philox streams): every tensor of both
    students' storages is bit-identical to the reference order (two rollouts one after the other on the same envs,
    adversarial_runner.py:709-733)."""
    monkeypatch.setenv("DCD_FUSED_ACT", fused)
    monkeypatch.setenv("DCD_HIP_GRAPHS", "0")
    got = {}
    for pair in ("1", "0"):
        monkeypatch.setenv("DCD_PAIR_ROLLOUT", pair)
        args = make_args(ued_algo="paired", num_processes=96, num_steps=40)
        runner, venv = build(args)
        assert runner.pair_rollouts == (pair == "1")
        torch.manual_seed(11)
        runner.run()                                                 # the storages keep the rollouts of this first update
        got[pair] = {name: {k: v.clone() for k, v in _storage_tensors(runner.agents[name].storage).items()}
                     for name in ("agent", "adversary_agent", "adversary_env")}
        venv.close()
    assert torch.equal(got["1"]["adversary_env"]["actions"], got["0"]["adversary_env"]["actions"])      # the same levels
    for name in ("agent", "adversary_agent"):
        assert got["1"][name]["actions"].float().std() > 0
        if fused == "0":
            # the modules draw from the ONE global torch generator: lock-step interleaves the two students' draws, so the
            # action streams differ from the sequential order (same distribution)
            assert torch.isfinite(got["1"][name]["returns"]).all()
            continue
        for k, v in got["1"][name].items():
            assert torch.equal(v, got["0"][name][k]), (name, k)


def test_paired_lockstep_training_runs_with_graphs_and_plr(monkeypatch):
    """The lock-step rollouts under HIP-graph replay and with level replay bookkeeping (REPAIRED-style: PLR + PAIRED,
    pair rollouts on the non-replay updates, sequential ones on replay updates): stats stay finite, both students
    learn, the buffer fills."""
    monkeypatch.setenv("DCD_PAIR_ROLLOUT", "1")
    args = make_args(ued_algo="paired", use_plr=True, level_replay_prob=0.5, level_replay_rho=0.1, level_replay_seed_buffer_size=200,
                     level_replay_strategy="positive_value_loss", level_replay_score_transform="rank", staleness_coef=0.3,
                     num_processes=64, num_steps=32, env_name="MultiGrid-MiniGoalLastAdversarial-v0")
    runner, venv = build(args)
    np.random.seed(1)
    kinds = set()
    for _ in range(8):
        s = runner.run()
        kinds.add(bool(runner.sampled_level_info["level_replay"]))
        assert np.isfinite(s["agent_value_loss"]) and np.isfinite(s["adversary_value_loss"])
    assert kinds == {True, False}
    assert any(k[0] == "pair" for k in runner._graphs)
    venv.close()


def test_replay_rollouts_from_per_step_graphs_equal_eager_rollouts(monkeypatch):
    """Robust PLR: a replay rollout re-samples a level for every finished episode on the host between two steps; its
    device part runs from one small HIP graph per step index (third replay rollout on).  Same seeds, DCD_HIP_GRAPHS=1 vs
    0: identical replay decisions, level seeds, sampler state and student storage after 14 updates.  (DCD_REPLAY_QUEUE=0: since
    round 4 the default takes finished episodes' levels from a device queue instead — tests/test_gpu_replay_queue.py — and this
    per-step path is what it falls back to.)"""
    res = {}
    monkeypatch.setenv("DCD_REPLAY_QUEUE", "0")
    for mode in ("1", "0"):
        monkeypatch.setenv("DCD_HIP_GRAPHS", mode)
        args = make_args(ued_algo="domain_randomization", use_plr=True, use_reset_random_dr=True, no_exploratory_grad_updates=True,
                         level_replay_strategy="positive_value_loss", level_replay_score_transform="rank", level_replay_prob=0.7,
                         level_replay_rho=0.1, level_replay_seed_buffer_size=80, staleness_coef=0.3, recurrent_adversary_env=False,
                         num_processes=64, num_steps=60, env_name="MultiGrid-MiniGoalLastAdversarial-v0")
        runner, venv = build(args)
        np.random.seed(5)
        torch.manual_seed(5)
        replays = []
        for _ in range(14):
            runner.run()
            replays.append(bool(runner.sampled_level_info["level_replay"]))
        s = runner.level_samplers["agent"]
        res[mode] = (replays, list(runner.current_level_seeds), s.seed_scores.copy(), s.seed_staleness.copy(), s.seeds.copy(),
                     {k: v.clone() for k, v in _storage_tensors(runner.agents["agent"].storage).items()},
                     any(isinstance(k, tuple) and k[0] == "student-steps" and isinstance(v, list) for k, v in runner._graphs.items()))
        venv.close()
    a, b = res["1"], res["0"]
    assert sum(a[0]) >= 4 and a[6] and not b[6]                 # the per-step graphs were captured and replayed
    assert a[0] == b[0] and a[1] == b[1]
    for i in (2, 3, 4):
        assert np.array_equal(a[i], b[i], equal_nan=True)
    for k in a[5]:
        assert torch.equal(a[5][k], b[5][k]), k


def test_rollout_return_stats_tail_from_a_suffix_of_envs():
    """_rollout_return_stats brings only a suffix of the env axis to the host: total episode count and the last-10 returns list
    (reference append order, adversarial_runner.py:845-853) equal the brute-force computation over the whole [T, B] buffers,
    also when episodes are so rare that the suffix has to widen."""
    import dcd_hip
    runner, venv = build(make_args(ued_algo="paired", num_processes=256, num_steps=64))
    runner.run()
    st = runner.agents["agent"].storage

    def brute():
        info, rets = st.ep_info.t().cpu().numpy(), st.ep_ret.t().cpu().numpy()
        b, t = np.nonzero(info & dcd_hip.INFO_EPISODE)
        order = np.lexsort((-t, b))
        return int(len(b)), rets[b[order], t[order]][-10:].tolist()
    got = runner._rollout_return_stats(st)
    assert (got["total_episodes"], got["returns_tail"]) == brute() and got["total_episodes"] >= 10
    keep = st.ep_info.clone()
    st.ep_info[:, 3:] &= ~dcd_hip.INFO_EPISODE                       # episodes only in the first three envs: the suffix must widen
    got = runner._rollout_return_stats(st)
    assert (got["total_episodes"], got["returns_tail"]) == brute()
    st.ep_info.zero_()
    got = runner._rollout_return_stats(st)
    assert got["total_episodes"] == 0 and got["returns_tail"] == []
    st.ep_info.copy_(keep)
    venv.close()
