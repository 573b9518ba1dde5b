"""Generate golden vectors by IMPORTING the reference's own modules (build container only).

Run once from the repo root:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
Needs /root/reference (absent on the GPU box: only the committed .npz files travel).

Importable there: algos.{storage,ppo,agent}, level_replay.{level_sampler,
level_store}, models.multigrid_models — with three in-process shims: numpy aliases removed in
numpy>=1.24 (np.float/np.bool/np.int/np.NINF), a stub `gym` module (storage.py imports it but
never uses it; models/walker_models.py needs gym.spaces.MultiDiscrete at import) and a stub
`lempel_ziv_complexity`.  envs/* is NOT importable (gym, gym_minigrid, baselines missing).

Only data is written: inputs + the reference's outputs.  No reference source is copied.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def install_shims():
    np.float = float
    np.bool = bool
    np.int = int
    np.NINF = -np.inf
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")

    class MultiDiscrete:  # only referenced by models/walker_models.py at import time
        pass

    spaces.MultiDiscrete = MultiDiscrete
    gym.spaces = spaces
    sys.modules["gym"] = gym
    sys.modules["gym.spaces"] = spaces
    lz = types.ModuleType("lempel_ziv_complexity")
    lz.lempel_ziv_complexity = lambda s: 0
    sys.modules["lempel_ziv_complexity"] = lz
    sys.path.insert(0, REF)


class Box:
    def __init__(self, shape, high=255):
        self.shape = tuple(shape)
        self.high = np.full(self.shape, high)


class Discrete:
    def __init__(self, n):
        self.n = n
        self.shape = ()

    def sample(self):
        return int(np.random.randint(self.n))


def student_spaces():
    return {"image": Box((3, 5, 5)), "direction": Box((1,), high=3)}, Discrete(7)


def synth_rollout(rng, T, B, A=7, p_done=0.12, p_trunc=0.3):
    """Synthetic buffers shaped like a student RolloutStorage after a rollout."""
    rewards = (rng.random((T, B, 1)) * (rng.random((T, B, 1)) < 0.05)).astype(np.float32)
    value_preds = rng.standard_normal((T + 1, B, 1)).astype(np.float32)
    masks = (rng.random((T + 1, B, 1)) > p_done).astype(np.float32)
    masks[0] = 1.0
    masks[T] = 0.0  # runner forces done on the last step
    bad = np.ones((T + 1, B, 1), np.float32)
    cliff = np.ones((T + 1, B, 1), np.float32)
    done_idx = masks == 0
    bad[done_idx & (rng.random((T + 1, B, 1)) < p_trunc)] = 0.0
    cliff[T][rng.random((B, 1)) < 0.5] = 0.0
    bad[T][cliff[T] == 0] = 0.0
    logits = rng.standard_normal((T, B, A)).astype(np.float32)
    logits = logits - np.log(np.exp(logits).sum(-1, keepdims=True))  # stored logits are normalised
    seeds = rng.integers(1, 12, (T, B, 1)).astype(np.int32)
    # a level seed is constant within an episode
    for b in range(B):
        cur = seeds[0, b, 0]
        for t in range(T):
            if t > 0 and masks[t, b, 0] == 0:
                cur = seeds[t, b, 0]
            seeds[t, b, 0] = cur
    return dict(rewards=rewards, value_preds=value_preds, masks=masks, bad_masks=bad,
                cliffhanger_masks=cliff, action_log_dist=logits, level_seeds=seeds)


def fill_storage(storage, d):
    storage.rewards.copy_(torch.from_numpy(d["rewards"]))
    storage.value_preds.copy_(torch.from_numpy(d["value_preds"]))
    storage.masks.copy_(torch.from_numpy(d["masks"]))
    storage.bad_masks.copy_(torch.from_numpy(d["bad_masks"]))
    storage.cliffhanger_masks.copy_(torch.from_numpy(d["cliffhanger_masks"]))
    storage.action_log_dist.copy_(torch.from_numpy(d["action_log_dist"]))
    storage.level_seeds.copy_(torch.from_numpy(d["level_seeds"]))


def gen_scans():
    from algos import RolloutStorage
    obs_space, act_space = student_spaces()
    out = {}
    for name, (T, B, seed) in {"small": (40, 8, 0), "paired": (256, 8, 1)}.items():
        rng = np.random.default_rng(seed)
        d = synth_rollout(rng, T, B)
        st = RolloutStorage(None, T, B, obs_space, act_space, 256, recurrent_arch="lstm")
        fill_storage(st, d)
        next_value = torch.from_numpy(rng.standard_normal((B, 1)).astype(np.float32))
        gamma, lam = 0.995, 0.95
        st.compute_returns(next_value, True, gamma, lam)
        gae = st.returns.clone().numpy()
        vp_after = st.value_preds.clone().numpy()
        st.returns.zero_()
        st.compute_returns(next_value, False, gamma, lam)
        disc = st.returns.clone().numpy()
        # value losses on the GAE returns
        st.returns.copy_(torch.from_numpy(gae))
        vl = {}
        for key, kw in {"abs": {}, "signed": dict(signed=True), "pos": dict(positive_only=True),
                        "abs_p2": dict(power=2), "abs_noclip": dict(clipped=False)}.items():
            vl[key] = st.get_batched_value_loss(**kw).numpy()
        for k, v in d.items():
            out[f"{name}_{k}"] = v
        out[f"{name}_next_value"] = next_value.numpy()
        out[f"{name}_value_preds_filled"] = vp_after
        out[f"{name}_gae"] = gae
        out[f"{name}_disc"] = disc
        for k, v in vl.items():
            out[f"{name}_vl_{k}"] = v
        out[f"{name}_gamma_lambda"] = np.array([gamma, lam])
    np.savez_compressed(os.path.join(OUT, "scans.npz"), **out)
    print("scans.npz", len(out))


class _Rollouts:
    """Minimal stand-in exposing the fields LevelSampler reads (level_sampler.py:490-525)."""

    def __init__(self, d, returns):
        t = torch.from_numpy
        self.level_seeds = t(d["level_seeds"])
        self.action_log_dist = t(d["action_log_dist"])
        self.masks = t(d["masks"])
        self.bad_masks = t(d["bad_masks"])
        self.cliffhanger_masks = t(d["cliffhanger_masks"])
        self.returns = t(returns)
        self.value_preds = t(d["value_preds"])
        self.rewards = t(d["rewards"])
        self.use_popart = False


def gen_sampler():
    from level_replay import LevelSampler
    obs_space, act_space = student_spaces()
    scans = np.load(os.path.join(OUT, "scans.npz"))
    d = {k[len("small_"):]: scans[k] for k in scans.files if k.startswith("small_")}
    T, B = d["rewards"].shape[:2]
    out = {}
    strategies = ["value_l1", "positive_value_loss", "signed_value_loss", "gae", "least_confidence",
                  "min_margin", "one_step_td_error", "uniform", "grounded_signed_value_loss",
                  "grounded_positive_value_loss"]
    for strat in strategies:
        # (a) fixed seed list ------------------------------------------------------------
        calls = []
        s = LevelSampler(list(range(1, 12)), obs_space, act_space, num_actors=B, strategy=strat,
                         score_transform="rank", temperature=0.1, staleness_coef=0.3, rho=0.5,
                         replay_prob=0.5, alpha=1.0, gamma=0.995)
        orig_full = s._partial_update_seed_score

        def log_full(actor_index, seed, score, max_score, num_steps, done=False, _o=orig_full):
            calls.append((actor_index, seed, float(score), float(max_score), num_steps, int(done)))
            return _o(actor_index, seed, score, max_score, num_steps, done=done)

        s._partial_update_seed_score = log_full
        ro = _Rollouts(d, d["gae"])
        s.update_with_rollouts(ro)
        out[f"{strat}_calls"] = np.array(calls, dtype=np.float64)
        out[f"{strat}_seed_scores"] = s.seed_scores.copy()
        out[f"{strat}_partial_scores"] = s.partial_seed_scores.copy()
        out[f"{strat}_partial_max"] = s.partial_seed_max_scores.copy()
        out[f"{strat}_partial_steps"] = s.partial_seed_steps.copy()
        out[f"{strat}_unseen"] = s.unseen_seed_weights.copy()
        if s.grounded_values is not None:
            out[f"{strat}_grounded"] = s.grounded_values.copy()
        out[f"{strat}_weights"] = s.sample_weights()
        s._partial_update_seed_score = orig_full
        s.after_update()
        out[f"{strat}_seed_scores_after"] = s.seed_scores.copy()
        out[f"{strat}_weights_after"] = s.sample_weights()
    # (b) score transforms on a fixed score vector -----------------------------------------
    rng = np.random.default_rng(5)
    scores = rng.random(11)
    stale = rng.integers(0, 30, 11).astype(float)
    unseen = (rng.random(11) < 0.25).astype(float)
    out["tf_scores"], out["tf_stale"], out["tf_unseen"] = scores, stale, unseen
    for tf in ["constant", "eps_greedy", "rank", "power", "softmax", "match", "match_rank"]:
        for sc in (0.0, 0.3):
            s = LevelSampler(list(range(1, 12)), obs_space, act_space, num_actors=B, strategy="value_l1",
                             score_transform=tf, temperature=0.3, staleness_coef=sc,
                             staleness_transform="power", staleness_temperature=1.0)
            s.seed_scores = scores.copy()
            s.seed_staleness = stale.copy()
            s.unseen_seed_weights = unseen.copy()
            out[f"tf_{tf}_{sc}"] = s.sample_weights()
    # (c) robust-PLR style buffer (sample_full_distribution) sequence ---------------------
    np.random.seed(123)
    s = LevelSampler([], obs_space, act_space, num_actors=B, strategy="positive_value_loss",
                     score_transform="rank", temperature=0.1, staleness_coef=0.3, rho=0.5,
                     replay_prob=0.5, alpha=1.0, sample_full_distribution=True, seed_buffer_size=6,
                     seed_buffer_priority="replay_support", gamma=0.995)
    log = []
    for it in range(4):
        rng = np.random.default_rng(100 + it)
        di = synth_rollout(rng, T, B)
        # staged seeds: new ids per iteration so the buffer overflows and evicts
        base = 1 + it * 5
        di["level_seeds"] = (di["level_seeds"] % 5 + base).astype(np.int32)
        ret = (di["value_preds"] + rng.standard_normal(di["value_preds"].shape).astype(np.float32) * 0.3)
        seeds_now = sorted(set(int(x) for x in di["level_seeds"].ravel()))
        solv = [bool(x % 2) for x in seeds_now]
        s.observe_external_unseen_sample(seeds_now, solv)
        s.update_with_rollouts(_Rollouts(di, ret))
        s.after_update()
        for k, v in di.items():
            out[f"buf{it}_{k}"] = v
        out[f"buf{it}_returns"] = ret
        out[f"buf{it}_observed"] = np.array(seeds_now)
        out[f"buf{it}_solvable"] = np.array(solv)
        out[f"buf{it}_seeds"] = s.seeds.copy()
        out[f"buf{it}_seed_scores"] = s.seed_scores.copy()
        out[f"buf{it}_staleness"] = s.seed_staleness.copy()
        out[f"buf{it}_unseen"] = s.unseen_seed_weights.copy()
        out[f"buf{it}_weights"] = s.sample_weights()
        out[f"buf{it}_solvable_mass"] = np.array(s.solvable_mass)
        out[f"buf{it}_filled"] = np.array(s._proportion_filled)
        log.append(len(s.staging_seed_set))
    out["buf_staging_sizes"] = np.array(log)
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **out)
    print("sampler.npz", len(out))


def gen_store():
    from level_replay import LevelStore
    out = {}
    st = LevelStore()
    a = st.insert(["1 2 3", "4 5 6", "1 2 3"])
    b = st.insert(["7 8 9", "4 5 6"], parent_seeds=[a[0], a[1]])
    out["ins_a"], out["ins_b"] = np.array(a), np.array(b)
    out["parents_len"] = np.array([len(st.seed2parent[k]) for k in sorted(st.seed2parent)])
    st.reconcile_seeds({1, 3})
    out["after_reconcile"] = np.array(sorted(st.seed2level))
    enc = np.arange(27, dtype=np.uint8).reshape(3, 3, 3)
    st2 = LevelStore(data_info={"numpy": True, "dtype": enc.dtype, "shape": enc.shape})
    sid = st2.insert([enc.tobytes()])
    out["bytes_roundtrip"] = st2.get_level(sid[0])
    np.savez_compressed(os.path.join(OUT, "store.npz"), **out)
    print("store.npz")


def gen_ppo():
    from algos import RolloutStorage, PPO, ACAgent
    from models import MultigridNetwork
    obs_space, act_space = student_spaces()
    torch.manual_seed(0)
    net = MultigridNetwork(observation_space=obs_space, action_space=act_space, scalar_fc=5, scalar_dim=4,
                           recurrent_arch="lstm", recurrent_hidden_size=256)
    out = {f"w_{k}": v.detach().numpy().copy() for k, v in net.state_dict().items()}
    out["n_params"] = np.array(sum(p.numel() for p in net.parameters()))
    T, B = 12, 4
    rng = np.random.default_rng(7)
    img = rng.integers(0, 11, (T + 1, B, 3, 5, 5)).astype(np.float32) / np.float32(10.0)
    direction = rng.integers(0, 4, (T + 1, B, 1)).astype(np.float32)
    masks = (rng.random((T + 1, B, 1)) > 0.3).astype(np.float32)
    masks[0] = 1
    masks[T] = 0
    bad = np.ones_like(masks)
    bad[(masks == 0) & (rng.random(masks.shape) < 0.8)] = 0
    bad[:, 0] = 1
    bad[5, 0] = 0          # env 0: exactly one truncated step (the reference skips it, storage.py:214-216)
    masks[5, 0] = 0
    rewards = (rng.random((T, B, 1)) * (rng.random((T, B, 1)) < 0.2)).astype(np.float32)
    algo = PPO(net, clip_param=0.2, ppo_epoch=2, num_mini_batch=1, value_loss_coef=0.5, entropy_coef=0.01,
               lr=1e-4, eps=1e-5, max_grad_norm=0.5, clip_value_loss=True, log_grad_norm=True)
    st = RolloutStorage(net, T, B, obs_space, act_space, 256, recurrent_arch="lstm",
                        use_proper_time_limits=True)
    agent = ACAgent(algo, st)
    # roll the policy over the fixed observation tape (actions sampled by the reference)
    st.obs["image"][0].copy_(torch.from_numpy(img[0]))
    st.obs["direction"][0].copy_(torch.from_numpy(direction[0]))
    trunc_img = rng.integers(0, 11, (T + 1, B, 3, 5, 5)).astype(np.float32) / np.float32(10.0)
    trunc_dir = rng.integers(0, 4, (T + 1, B, 1)).astype(np.float32)
    st.truncated_obs["image"].copy_(torch.from_numpy(trunc_img))
    st.truncated_obs["direction"].copy_(torch.from_numpy(trunc_dir))
    torch.manual_seed(1)
    vals, acts, dists = [], [], []
    for t in range(T):
        with torch.no_grad():
            v, a, ld, h = agent.act(st.get_obs(t), st.get_recurrent_hidden_state(t), st.masks[t])
        lp = ld.gather(-1, a)
        vals.append(v.numpy().copy())
        acts.append(a.numpy().copy())
        dists.append(ld.numpy().copy())
        obs = {"image": torch.from_numpy(img[t + 1]), "direction": torch.from_numpy(direction[t + 1])}
        agent.insert(obs, h, a, lp, ld, v, torch.from_numpy(rewards[t]), torch.from_numpy(masks[t + 1]),
                     torch.from_numpy(bad[t + 1]))
    with torch.no_grad():
        nv = agent.get_value(st.get_obs(-1), st.get_recurrent_hidden_state(-1), st.masks[-1])
    st.compute_returns(nv, True, 0.995, 0.95)
    out.update(img=img, direction=direction, masks=masks, bad_masks=bad, rewards=rewards,
               trunc_img=trunc_img, trunc_dir=trunc_dir, act_values=np.stack(vals), act_actions=np.stack(acts),
               act_logdist=np.stack(dists), hidden=st.recurrent_hidden_states.numpy().copy(),
               next_value=nv.numpy().copy(), truncated_value_preds=st.truncated_value_preds.numpy().copy(),
               returns=st.returns.numpy().copy(), action_log_probs=st.action_log_probs.numpy().copy())
    # evaluate_actions over the whole tape (sequence mode of the masked RNN)
    with torch.no_grad():
        flat = lambda x: x[:-1].reshape(T * B, *x.shape[2:])
        ev, elp, eent, _ = net.evaluate_actions(
            {"image": flat(st.obs["image"]), "direction": flat(st.obs["direction"])},
            st._split_batched_lstm_recurrent_hidden_states(st.recurrent_hidden_states[0]),
            flat(st.masks), st.actions.reshape(T * B, 1))
    out.update(eval_values=ev.numpy().copy(), eval_logp=elp.numpy().copy(), eval_entropy=np.array(eent.item()))
    torch.manual_seed(2)
    vl, al, ent, info = agent.update()
    out.update(ppo_value_loss=np.array(vl), ppo_action_loss=np.array(al), ppo_entropy=np.array(ent),
               ppo_grad_norms=np.array(info["grad_norms"]))
    for k, v in net.state_dict().items():
        out[f"w_after_{k}"] = v.detach().numpy().copy()
    # discard_grad path: parameters must not move
    torch.manual_seed(3)
    st.compute_returns(nv, True, 0.995, 0.95)
    vl2, al2, ent2, _ = agent.update(discard_grad=True)
    out.update(ppo2_value_loss=np.array(vl2), ppo2_action_loss=np.array(al2), ppo2_entropy=np.array(ent2))
    np.savez_compressed(os.path.join(OUT, "ppo.npz"), **out)
    print("ppo.npz", len(out))

    # teacher net shape facts only (22.5M params: weights not stored)
    adv_space = {"image": Box((3, 15, 15)), "time_step": Box((1,), high=27), "random_z": Box((50,), high=1)}
    torch.manual_seed(0)
    tnet = MultigridNetwork(observation_space=adv_space, action_space=Discrete(169), conv_filters=128,
                            scalar_fc=10, scalar_dim=28, random_z_dim=50, recurrent_arch="lstm",
                            recurrent_hidden_size=256)
    shapes = {k: np.array(v.shape) for k, v in tnet.state_dict().items()}
    shapes["n_params"] = np.array(sum(p.numel() for p in tnet.parameters()))
    np.savez_compressed(os.path.join(OUT, "teacher_shapes.npz"), **shapes)
    print("teacher_shapes.npz")


def _roll_and_update(net, st, agent, tape, T, masks, bad, rewards, out, seeds=(1, 2)):
    """Rolls the REFERENCE policy over a fixed observation tape (its own sampled actions are stored; they do not feed back into
    the tape), computes returns with proper time limits and runs ONE reference PPO update.  tape(t) -> obs dict of step t."""
    for k, v in tape(0).items():
        st.obs[k][0].copy_(v)
    torch.manual_seed(seeds[0])
    vals, acts, dists = [], [], []
    for t in range(T):
        with torch.no_grad():
            v, a, ld, h = agent.act(st.get_obs(t), st.get_recurrent_hidden_state(t), st.masks[t])
        lp = ld.gather(-1, a)
        vals.append(v.numpy().copy()); acts.append(a.numpy().copy()); dists.append(ld.numpy().copy())
        agent.insert(tape(t + 1), h, a, lp, ld, v, torch.from_numpy(rewards[t]), torch.from_numpy(masks[t + 1]),
                     torch.from_numpy(bad[t + 1]))
    with torch.no_grad():
        nv = agent.get_value(st.get_obs(-1), st.get_recurrent_hidden_state(-1), st.masks[-1])
    st.compute_returns(nv, True, 0.995, 0.95)
    out.update(masks=masks, bad_masks=bad, rewards=rewards, act_values=np.stack(vals), act_actions=np.stack(acts),
               act_logdist=np.stack(dists), next_value=nv.numpy().copy(), returns=st.returns.numpy().copy(),
               action_log_probs=st.action_log_probs.numpy().copy(), hidden0=st.recurrent_hidden_states[0].numpy().copy(),
               hidden1=st.recurrent_hidden_states[1].numpy().copy(), hiddenT=st.recurrent_hidden_states[T].numpy().copy())
    if st.truncated_value_preds is not None:
        out["truncated_value_preds"] = st.truncated_value_preds.numpy().copy()
    agent64 = _as_double(agent)
    torch.manual_seed(seeds[1])
    vl, al, ent, info = agent.update()
    out.update(ppo_value_loss=np.array(vl), ppo_action_loss=np.array(al), ppo_entropy=np.array(ent),
               ppo_grad_norms=np.array(info["grad_norms"]))
    # the SAME update by the same reference code evaluated in float64 (same inputs: the fp32 rollout above, cast): what the
    # expressions of ppo.py:61-146 give without fp32 round-off.  The reference's own fp32 CPU run sits up to 8e-4 (gradient
    # norm of the 22.5 M-parameter teacher) away from it; the device path is compared with both.
    torch.manual_seed(seeds[1])
    vl, al, ent, info = agent64.update()
    out.update(f64_ppo_value_loss=np.array(vl), f64_ppo_action_loss=np.array(al), f64_ppo_entropy=np.array(ent),
               f64_ppo_grad_norms=np.array(info["grad_norms"]))
    return agent64


def _as_double(agent):
    """A deep copy of a reference ACAgent (network, PPO with its optimiser, RolloutStorage) cast to float64.  The reference's
    `one_hot` returns float32 (models/common.py:211-213): a forward pre-hook on the embedding that consumes it casts its input
    to the weights' dtype — the reference's functions themselves are untouched."""
    import copy
    a = copy.deepcopy(agent)
    net = a.algo.actor_critic
    net.double()
    net.scalar_embed.register_forward_pre_hook(lambda m, inp: (inp[0].to(m.weight.dtype),))
    st = a.storage
    for name, val in list(vars(st).items()):
        if isinstance(val, torch.Tensor) and val.dtype == torch.float32:
            setattr(st, name, val.double())
        elif isinstance(val, dict):
            setattr(st, name, {k: (v.double() if isinstance(v, torch.Tensor) and v.dtype == torch.float32 else v) for k, v in val.items()})
    return a


def gen_ppo_wide():
    """Student PPO at a width where the GPU product path's large-batch branches engage when their row thresholds are lowered
    (split-K weight gradients need >= 512 rows per minibatch: T = 32, B = 32, two minibatches of 16 envs).  Initial weights:
    the reference's own seed-0 initialisation already held in ppo.npz.  Observations are stored as the engine's raw bytes
    ([vx][vy][c] + direction); the reference sees them as VecPreprocessImageWrapper would hand them over (k / 10, CHW)."""
    from algos import RolloutStorage, PPO, ACAgent
    from models import MultigridNetwork
    obs_space, act_space = student_spaces()
    g0 = np.load(os.path.join(OUT, "ppo.npz"))
    net = MultigridNetwork(observation_space=obs_space, action_space=act_space, scalar_fc=5, scalar_dim=4,
                           recurrent_arch="lstm", recurrent_hidden_size=256)
    net.load_state_dict({k[2:]: torch.from_numpy(g0[k]) for k in g0.files if k.startswith("w_") and not k.startswith("w_after_")})
    T, B = 32, 32
    rng = np.random.default_rng(11)
    img8 = rng.integers(0, 11, (T + 1, B, 5, 5, 3)).astype(np.uint8)
    dir8 = rng.integers(0, 4, (T + 1, B)).astype(np.uint8)
    trunc_img8 = rng.integers(0, 11, (T + 1, B, 5, 5, 3)).astype(np.uint8)
    trunc_dir8 = rng.integers(0, 4, (T + 1, B)).astype(np.uint8)
    f32 = lambda im: torch.from_numpy((im.astype(np.float64) / 10.0).astype(np.float32).transpose(*range(im.ndim - 3), -1, -3, -2).copy())
    masks = (rng.random((T + 1, B, 1)) > 0.08).astype(np.float32)
    masks[0] = 1
    masks[T] = 0
    bad = np.ones_like(masks)
    bad[(masks == 0) & (rng.random(masks.shape) < 0.6)] = 0
    rewards = (rng.random((T, B, 1)) * (rng.random((T, B, 1)) < 0.1)).astype(np.float32)
    algo = PPO(net, clip_param=0.2, ppo_epoch=2, num_mini_batch=2, value_loss_coef=0.5, entropy_coef=0.0,
               lr=1e-4, eps=1e-5, max_grad_norm=0.5, clip_value_loss=True, log_grad_norm=True)
    st = RolloutStorage(net, T, B, obs_space, act_space, 256, recurrent_arch="lstm", use_proper_time_limits=True)
    agent = ACAgent(algo, st)
    st.truncated_obs["image"].copy_(f32(trunc_img8))
    st.truncated_obs["direction"].copy_(torch.from_numpy(trunc_dir8.astype(np.float32))[..., None])
    tape = lambda t: {"image": f32(img8[t]), "direction": torch.from_numpy(dir8[t].astype(np.float32))[:, None]}
    out = dict(img8=img8, dir8=dir8, trunc_img8=trunc_img8, trunc_dir8=trunc_dir8, T=np.array(T), B=np.array(B))
    a64 = _roll_and_update(net, st, agent, tape, T, masks, bad, rewards, out)
    for k, v in net.state_dict().items():
        out[f"w_after_{k}"] = v.detach().numpy().copy()
    for k, v in a64.algo.actor_critic.state_dict().items():
        out[f"f64_w_after_{k}"] = v.detach().numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "ppo_wide.npz"), **out)
    print("ppo_wide.npz", len(out), "grad norms", out["ppo_grad_norms"], "float64:", out["f64_ppo_grad_norms"], out["ppo_value_loss"], out["f64_ppo_value_loss"])


def gen_ppo_teacher():
    """The PAIRED teacher (recurrent_adversary_env: conv 3 -> 128 on the 15 x 15 grid, 21 692 -> LSTM 256, 169 actions;
    22 502 732 parameters) over a constructive tape of T = 27 steps, B = 8 envs: per-step act outputs and ONE reference PPO
    update (2 epochs, 1 minibatch).  90 MB of weights are not stored: the initial weights are `det_weights.fill_state_dict`
    (integer hash, identical on every host); after the update every tensor but the LSTM input weight is stored whole, and of
    that one a column sample (every 89th image column + the 60 non-image columns) plus the fp64 row / column sums of its
    change."""
    sys.path.insert(0, OUT)
    import det_weights
    from algos import RolloutStorage, PPO, ACAgent
    from models import MultigridNetwork
    adv_space = {"image": Box((3, 15, 15)), "time_step": Box((1,), high=27), "random_z": Box((50,), high=1)}
    net = MultigridNetwork(observation_space=adv_space, action_space=Discrete(169), conv_filters=128, scalar_fc=10, scalar_dim=28,
                           random_z_dim=50, recurrent_arch="lstm", recurrent_hidden_size=256)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    w0 = det_weights.fill_state_dict(shapes, gains={"actor.1.linear.weight": 0.01})
    net.load_state_dict({k: torch.from_numpy(v) for k, v in w0.items()})
    T, B = 27, 8
    rng = np.random.default_rng(5)
    enc, loc = det_weights.teacher_tape(rng, T, B)
    z = rng.random((T + 1, B, 50)).astype(np.float32)
    masks = np.ones((T + 1, B, 1), np.float32)
    masks[T] = 0
    masks[9, 2] = masks[17, 5] = 0                     # the recurrent state is cut mid-tape somewhere (the runner never does)
    bad = np.ones_like(masks)
    rewards = np.zeros((T, B, 1), np.float32)
    rewards[-1] = rng.random((B, 1)).astype(np.float32)          # replace_final_return: the regret arrives at the last step
    f32 = lambda im: torch.from_numpy((im.astype(np.float64) / 10.0).astype(np.float32).transpose(0, 3, 1, 2).copy())
    tape = lambda t: {"image": f32(enc[t]), "time_step": torch.full((B, 1), float(t)), "random_z": torch.from_numpy(z[t])}
    algo = PPO(net, clip_param=0.2, ppo_epoch=2, num_mini_batch=1, value_loss_coef=0.5, entropy_coef=0.0,
               lr=1e-4, eps=1e-5, max_grad_norm=0.5, clip_value_loss=True, log_grad_norm=True)
    st = RolloutStorage(net, T, B, adv_space, Discrete(169), 256, recurrent_arch="lstm", use_proper_time_limits=False)
    agent = ACAgent(algo, st)
    out = dict(enc=enc, loc=loc, random_z=z, T=np.array(T), B=np.array(B))
    a64 = _roll_and_update(net, st, agent, tape, T, masks, bad, rewards, out, seeds=(4, 6))
    big = "rnn.rnn.weight_ih_l0"
    cols = np.concatenate([np.arange(0, 21632, 89), np.arange(21632, 21692)])
    out["big_cols"] = cols
    for pre, sd in (("", net.state_dict()), ("f64_", a64.algo.actor_critic.state_dict())):
        for k, v in sd.items():
            v = v.detach().numpy()
            if k == big:
                d = v.astype(np.float64) - w0[k].astype(np.float64)
                out[pre + "w_after_big_sample"] = v[:, cols].astype(np.float32)
                out[pre + "big_delta_rowsum"] = d.sum(1)
                out[pre + "big_delta_colsum"] = d.sum(0)
                out[pre + "big_delta_abs_mean"] = np.array(np.abs(d).mean())
            else:
                out[f"{pre}w_after_{k}"] = v.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "ppo_teacher.npz"), **out)
    print("ppo_teacher.npz", len(out), "losses", out["ppo_value_loss"], out["ppo_action_loss"], out["ppo_entropy"], "grad norms", out["ppo_grad_norms"])
    print("   float64:", out["f64_ppo_value_loss"], out["f64_ppo_action_loss"], out["f64_ppo_entropy"], out["f64_ppo_grad_norms"])


def gen_entropy():
    """policy_entropy (level_sampler.py:278-287): `-exp(l) * l.sum(-1) / max_entropy` broadcasts [L,A] against [L], so
    the reference's expression is defined only for episodes of L == 1 or L == A steps (any other length raises).
    Rollouts whose every segment has one of those two lengths, through the reference's LevelSampler; plus direct calls
    of _average_entropy on single episodes."""
    from level_replay import LevelSampler
    obs_space, act_space = student_spaces()
    A, T, B = 7, 31, 8
    rng = np.random.default_rng(11)
    d = synth_rollout(rng, T, B, A=A)
    masks = np.ones((T + 1, B, 1), np.float32)
    for b in range(B):
        t = 0
        while t < T:
            left = T - t
            L = 7 if (left >= 7 and rng.random() < 0.6) else 1
            t += L
            masks[t, b, 0] = 0.0                      # done closes [start, t)
        if b % 3 == 2:
            masks[T, b, 0] = 1.0                      # trailing PARTIAL segment (still 1 or 7 steps long)
    d["masks"] = masks
    cliff = np.ones_like(masks)
    cliff[T][(masks[T] == 0) & (rng.random((B, 1)) < 0.4)] = 0.0
    d["cliffhanger_masks"] = cliff
    seeds = d["level_seeds"]
    for b in range(B):                                 # a level seed is constant within an episode
        cur = seeds[0, b, 0]
        for t in range(T):
            if t > 0 and masks[t, b, 0] == 0:
                cur = int(rng.integers(1, 12))
            seeds[t, b, 0] = cur
    out = {f"roll_{k}": v for k, v in d.items()}
    calls = []
    s = LevelSampler(list(range(1, 12)), obs_space, act_space, num_actors=B, strategy="policy_entropy",
                     score_transform="rank", temperature=0.1, staleness_coef=0.3, rho=0.5, replay_prob=0.5, alpha=1.0)
    orig = s._partial_update_seed_score

    def log(actor_index, seed, score, max_score, num_steps, done=False, _o=orig):
        calls.append((actor_index, seed, float(score), float(max_score), num_steps, int(done)))
        return _o(actor_index, seed, score, max_score, num_steps, done=done)

    s._partial_update_seed_score = log
    s.update_with_rollouts(_Rollouts(d, d["value_preds"]))
    out["calls"] = np.array(calls, dtype=np.float64)
    out["seed_scores"] = s.seed_scores.copy()
    # direct calls on normalised log-probabilities
    for L in (1, 7):
        l = torch.log_softmax(torch.from_numpy(rng.standard_normal((L, A)).astype(np.float32) * 2), -1)
        out[f"direct{L}_logits"] = l.numpy().copy()
        out[f"direct{L}_mean_max"] = np.array(s._average_entropy(episode_logits=l))
    np.savez_compressed(os.path.join(OUT, "entropy.npz"), **out)
    print("entropy.npz", len(out), "calls", len(calls))


def gen_staged():
    """Rollouts at the widths the LDS-staged scoring kernels take (B a multiple of 32; one and two time tiles): T = 256, B = 64
    and T = 300, B = 96, two rollouts per shape (PAIRED's two students).  Returns through the reference's
    RolloutStorage.compute_returns (storage.py:233-256), then the (actor, seed, score, max, steps, done) sequence its LevelSampler
    passes to _partial_update_seed_score (level_sampler.py:486-578) and the sampler state after it, for the value strategies.
    The logits buffer is all zeros (these strategies ignore it) and is not stored."""
    from algos import RolloutStorage
    from level_replay import LevelSampler
    obs_space, act_space = student_spaces()
    gamma, lam = 0.995, 0.95
    out = {"gamma_lambda": np.array([gamma, lam])}
    strategies = ["positive_value_loss", "value_l1", "signed_value_loss", "one_step_td_error", "grounded_positive_value_loss",
                  "grounded_signed_value_loss"]
    for name, (T, B, seed) in {"a0": (256, 64, 31), "a1": (256, 64, 32), "b0": (300, 96, 33), "b1": (300, 96, 34)}.items():
        rng = np.random.default_rng(seed)
        d = synth_rollout(rng, T, B, p_done=0.03)
        d["action_log_dist"] = np.zeros((T, B, 7), np.float32)
        if name.endswith("1"):                           # second student: some envs end the rollout mid-episode
            open_end = rng.random((B, 1)) < 0.3
            d["masks"][T][open_end] = 1.0
            d["cliffhanger_masks"][T][open_end] = 1.0
        st = RolloutStorage(None, T, B, obs_space, act_space, 256, recurrent_arch="lstm")
        fill_storage(st, d)
        next_value = torch.from_numpy(rng.standard_normal((B, 1)).astype(np.float32))
        st.compute_returns(next_value, True, gamma, lam)
        gae = st.returns.clone().numpy()
        d["value_preds"] = st.value_preds.clone().numpy()      # row T now holds next_value (storage.py:236)
        for k in ("rewards", "value_preds", "masks", "cliffhanger_masks", "level_seeds"):
            out[f"{name}_{k}"] = d[k][..., 0]
        out[f"{name}_gae"] = gae[..., 0]
        for strat in strategies:
            calls = []
            s = LevelSampler(list(range(1, 12)), obs_space, act_space, num_actors=B, strategy=strat, score_transform="rank",
                             temperature=0.1, staleness_coef=0.3, rho=0.5, replay_prob=0.5, alpha=1.0, gamma=gamma)
            orig = s._partial_update_seed_score

            def log(actor_index, seed, score, max_score, num_steps, done=False, _o=orig):
                calls.append((actor_index, seed, float(score), float(max_score), num_steps, int(done)))
                return _o(actor_index, seed, score, max_score, num_steps, done=done)

            s._partial_update_seed_score = log
            s.update_with_rollouts(_Rollouts(d, gae))
            c = np.array(calls, dtype=np.float64)
            out[f"{name}_{strat}_calls_i"] = c[:, [0, 1, 4, 5]].astype(np.int32)
            out[f"{name}_{strat}_calls_f"] = c[:, [2, 3]]
            out[f"{name}_{strat}_seed_scores"] = s.seed_scores.copy()
            if s.grounded_values is not None:
                out[f"{name}_{strat}_grounded"] = s.grounded_values.copy()
    np.savez_compressed(os.path.join(OUT, "staged.npz"), **out)
    print("staged.npz", len(out), {k: len(out[f"{k}_positive_value_loss_calls_i"]) for k in ("a0", "a1", "b0", "b1")})


def gen_popart():
    """models/popart.py:20-103 and its use by RolloutStorage (algos/storage.py:240-256,301-306): moments after three
    updates, normalize / denormalize / forward, GAE returns and batched value loss on denormalised value predictions."""
    from models.popart import PopArt
    from algos import RolloutStorage
    torch.manual_seed(4)
    pa = PopArt(6, 1)
    out = {"w0": pa.weight.detach().numpy().copy(), "b0": pa.bias.detach().numpy().copy()}
    rng = np.random.default_rng(21)
    for i in range(3):
        x = (rng.standard_normal((64, 1)) * (1 + 2 * i) + 3 * i).astype(np.float32)
        pa.update(torch.from_numpy(x))
        out[f"upd{i}_x"] = x
        for k in ("weight", "bias", "stddev", "mean", "mean_sq", "debiasing_term"):
            out[f"upd{i}_{k}"] = getattr(pa, k).detach().numpy().copy()
    q = rng.standard_normal((10, 1)).astype(np.float32)
    feats = rng.standard_normal((10, 6)).astype(np.float32)
    out.update(q=q, feats=feats, normalize=pa.normalize(torch.from_numpy(q)).detach().numpy(),
               denormalize=pa.denormalize(torch.from_numpy(q)).detach().numpy(),
               forward=pa(torch.from_numpy(feats)).detach().numpy())
    obs_space, act_space = student_spaces()
    T, B = 40, 8
    d = synth_rollout(np.random.default_rng(0), T, B)

    class _Model:
        popart = pa
    st = RolloutStorage(_Model(), T, B, obs_space, act_space, 256, recurrent_arch="lstm", use_popart=True)
    fill_storage(st, d)
    nv = torch.from_numpy(rng.standard_normal((B, 1)).astype(np.float32))
    st.compute_returns(nv, True, 0.995, 0.95)
    out.update({f"roll_{k}": v for k, v in d.items()})
    out.update(next_value=nv.numpy(), gae=st.returns.numpy().copy(), denorm=st.denorm_value_preds.detach().numpy().copy(),
               vl_abs=st.get_batched_value_loss().detach().numpy(), vl_pos=st.get_batched_value_loss(positive_only=True).detach().numpy())
    np.savez_compressed(os.path.join(OUT, "popart.npz"), **out)
    print("popart.npz", len(out))


def gen_bezier():
    """envs/box2d/bezier.py:get_bezier_curve with the arguments CarRacingBezier passes (car_racing_bezier.py:297-303):
    random control-point sets of 3..12 points (the env's own generator, get_random_points, scaled to the playfield) and
    adversary-style sets on the 10 x 10 sketch grid (car_racing_adversarial.py:236-244)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_bezier", os.path.join(REF, "envs", "box2d", "bezier.py"))
    bz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bz)
    rng = np.random.RandomState(3)
    out, k = {}, 0
    for n in (3, 4, 5, 7, 9, 12, 12, 12):
        a = bz.get_random_points(n=n, scale=2000 / 6.0, np_random=rng)
        x, y, ctrl = bz.get_bezier_curve(a=a, rad=0.2, edgy=0.2, numpoints=40)
        out[f"c{k}_pts"], out[f"c{k}_x"], out[f"c{k}_y"], out[f"c{k}_a"] = a, x, y, ctrl
        k += 1
    ratio = (2000 / 6.0) / 10
    for n in (5, 8, 12, 12):                       # distinct cells of the adversary's sketch, un-normalised as the env does
        cells = rng.permutation(100)[:n]
        a = np.array([((c % 10 + 1) * ratio, (c // 10 + 1) * ratio) for c in cells], dtype=np.float64)
        x, y, ctrl = bz.get_bezier_curve(a=a, rad=0.2, edgy=0.2, numpoints=40)
        out[f"c{k}_pts"], out[f"c{k}_x"], out[f"c{k}_y"], out[f"c{k}_a"] = a, x, y, ctrl
        k += 1
    out["n_cases"] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "bezier.npz"), **out)
    print("bezier.npz", k, "curves")


def _digest(p):
    """shape, fp64 sum, fp64 sum of squares and the first 8 entries of a tensor (keeps the fixture small)."""
    a = p.detach().numpy().astype(np.float64).ravel()
    return np.concatenate([np.array(p.shape, np.float64), [a.sum(), (a * a).sum()], a[:8]])


def gen_f4_nets():
    """models/car_racing_models.py:18-165 and models/walker_models.py:113-167,345-376: initial weights from a fixed seed and
    evaluate_actions / get_value on fixed inputs (deterministic: no sampling)."""
    from models.car_racing_models import CarRacingNetwork
    from models.walker_models import BipedalWalkerStudentPolicy
    rng = np.random.default_rng(12)
    out = {}
    act3 = Box((3,))
    act3.low, act3.high = np.array([-1.0, 0.0, 0.0]), np.array([1.0, 1.0, 1.0])
    for tag, kw, hw in (("cr", dict(), 96), ("crpop", dict(use_popart=True), 96), ("crcrop", dict(crop=True), 84)):
        torch.manual_seed(0)
        net = CarRacingNetwork((4, hw, hw), act3, **kw)
        x = rng.random((2, 4, hw, hw)).astype(np.float32)
        a = rng.random((2, 3)).astype(np.float32) * 0.98 + 0.01
        with torch.no_grad():
            v, lp, ent, _ = net.evaluate_actions(torch.from_numpy(x), None, None, torch.from_numpy(a))
        out.update({f"{tag}_w_{k}": _digest(p) for k, p in net.state_dict().items()})     # 0.4M weights per net: digests
        out.update({f"{tag}_x": x, f"{tag}_a": a, f"{tag}_value": v.numpy(), f"{tag}_logp": lp.numpy(), f"{tag}_ent": np.array(ent.item()),
                    f"{tag}_proc": net.process_action(a)})
    act4 = Box((4,))
    torch.manual_seed(0)
    net = BipedalWalkerStudentPolicy((24,), act4)
    x = rng.standard_normal((7, 24)).astype(np.float32)
    a = rng.standard_normal((7, 4)).astype(np.float32)
    with torch.no_grad():
        v, lp, ent, _ = net.evaluate_actions(torch.from_numpy(x), None, None, torch.from_numpy(a))
        det = net.act(torch.from_numpy(x), None, None, deterministic=True)
    out.update({f"bw_w_{k}": p.detach().numpy().copy() for k, p in net.state_dict().items()})
    out.update(bw_x=x, bw_a=a, bw_value=v.numpy(), bw_logp=lp.numpy(), bw_ent=np.array(ent.item()), bw_mode=det[1].numpy(), bw_mode_logp=det[2].numpy())
    np.savez_compressed(os.path.join(OUT, "f4_nets.npz"), **out)
    print("f4_nets.npz", len(out))


def gen_f4_adv():
    """models/walker_models.py:170-257 BipedalWalkerAdversaryPolicy (the level builder of BASELINE config 4; the continuous,
    non-editor form every shipped config uses): initial weights from a fixed seed, deterministic act (incl. the reference's
    log-prob of the SQUASHED action), get_value, evaluate_actions on fixed inputs."""
    from models.walker_models import BipedalWalkerAdversaryPolicy
    rng = np.random.default_rng(21)
    obs = {"image": Box((8,)), "random_z": Box((10,)), "time_step": Box((1,))}
    act = Box((8,))
    torch.manual_seed(0)
    net = BipedalWalkerAdversaryPolicy(obs, act)
    x = {"image": rng.standard_normal((5, 8)).astype(np.float32), "random_z": rng.random((5, 10)).astype(np.float32),
         "time_step": np.arange(5, dtype=np.float32).reshape(5, 1)}
    a = np.tanh(rng.standard_normal((5, 8))).astype(np.float32)
    xt = {k: torch.from_numpy(v) for k, v in x.items()}
    with torch.no_grad():
        v, lp, ent, _ = net.evaluate_actions(xt, None, None, torch.from_numpy(a))
        det = net.act(xt, None, None, deterministic=True)
        gv = net.get_value(xt, None, None)
    out = {f"w_{k}": p.detach().numpy().copy() for k, p in net.state_dict().items()}
    out.update({f"x_{k}": v_ for k, v_ in x.items()})
    out.update(a=a, value=v.numpy(), logp=lp.numpy(), ent=np.array(ent.item()), det_value=det[0].numpy(), det_action=det[1].numpy(),
               det_logp=det[2].numpy(), get_value=gv.numpy())
    np.savez_compressed(os.path.join(OUT, "f4_adv.npz"), **out)
    print("f4_adv.npz", len(out))


def gen_f4_cr_adv():
    """models/car_racing_models.py:168-530 CarRacingBezierAdversaryEnvNetwork (the level builder of BASELINE config 5), spaces as
    envs/box2d/car_racing_adversarial.py:58-127 builds them (10 x 10 sketch, 12 control points, random_z 50): initial weights from
    a fixed seed (digests), then get_value / evaluate_actions on fixed inputs and act under a fixed torch seed, for the continuous
    (Beta) form, the categorical form with the skip action, and the sparse-reward form with a goal-bin step (+ PopArt)."""
    from models.car_racing_models import CarRacingBezierAdversaryEnvNetwork
    rng = np.random.default_rng(33)
    out = {}
    cases = {"beta": (dict(), 12, False), "cat": (dict(use_categorical=True, use_skip=True), 12, False),
             "goal": (dict(use_categorical=True, use_skip=True, use_goal=True, num_goal_bins=24, use_popart=True), 13, True)}
    for tag, (kw, max_steps, goal) in cases.items():
        obs = {"control_points": Box((1, 10, 10), high=10), "time_step": Box((1,), high=max_steps), "random_z": Box((50,), high=1)}
        if goal:
            obs["goal_bin"] = Box((1,), high=25)
        if kw.get("use_categorical"):
            act = Box((2,) if goal else (1,))
            act.high = np.array((1, 101)) if goal else np.array((101,))
        else:
            act = Box((3,), high=1)
        torch.manual_seed(0)
        net = CarRacingBezierAdversaryEnvNetwork(obs, act, **kw)
        out.update({f"{tag}_w_{k}": _digest(p) for k, p in net.state_dict().items()})
        B = 6
        sketch = (rng.random((B, 1, 10, 10)) < 0.06).astype(np.float32)
        sketch[0] = 0                                                      # an empty sketch: fewer than 3 points, skip masked
        ts = rng.integers(0, max_steps - 1, (B, 1)).astype(np.float32)
        if goal:
            ts[4] = max_steps - 1                                          # ONE goal step in the batch (time_step == total - 1)
        x = {"control_points": sketch, "time_step": ts, "random_z": rng.random((B, 50)).astype(np.float32)}
        if goal:
            x["goal_bin"] = np.full((B, 1), 24, np.float32)
        xt = {k: torch.from_numpy(v) for k, v in x.items()}
        if kw.get("use_categorical"):
            free = [np.flatnonzero(sketch[b].ravel() == 0) + 1 for b in range(B)]
            a = np.array([[float(rng.choice(f))] for f in free], np.float32)
            if goal:
                a[4] = 7.0
                a = np.concatenate([(ts == max_steps - 1).astype(np.float32), a], axis=1)
        else:
            a = (rng.random((B, 3)) * 0.96 + 0.02).astype(np.float32)
        with torch.no_grad():
            v, lp, ent, _ = net.evaluate_actions(xt, None, None, torch.from_numpy(a))
            gv = net.get_value(xt, None, None)
            # act: every row of a batch is at the same step (it reads time_step[0]); a fixed torch seed fixes the draw
            one = {k: t.clone() for k, t in xt.items()}
            one["time_step"][:] = 3.0
            torch.manual_seed(5)
            av, aa, alp, _ = net.act(one, None, None)
            out.update({f"{tag}_act_value": av.numpy(), f"{tag}_act_action": aa.numpy().astype(np.float64), f"{tag}_act_logp": alp.numpy()})
            if goal:
                one["time_step"][:] = float(max_steps - 1)
                torch.manual_seed(6)
                gvv, ga, glp, _ = net.act(one, None, None)
                out.update({f"{tag}_gact_action": ga.numpy().astype(np.float64), f"{tag}_gact_logp": glp.numpy()})
                out[f"{tag}_proc_goal"] = net.process_action(ga).numpy().astype(np.float64)
            if kw.get("use_categorical"):
                out[f"{tag}_proc"] = net.process_action(aa).numpy().astype(np.float64)
        out.update({f"{tag}_x_{k}": v_ for k, v_ in x.items()})
        out.update({f"{tag}_a": a, f"{tag}_value": v.numpy(), f"{tag}_logp": lp.numpy(), f"{tag}_ent": np.array(float(ent)),
                    f"{tag}_get_value": gv.numpy()})
    np.savez_compressed(os.path.join(OUT, "f4_cr_adv.npz"), **out)
    print("f4_cr_adv.npz", len(out))


if __name__ == "__main__":
    import numpy.ma  # noqa: F401  (must be imported BEFORE the alias shims: numpy 2 uses np.bool as a dtype there)
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot  # noqa: F401  (envs/box2d/bezier.py imports it at module level)
    install_shims()
    torch.set_num_threads(1)
    todo = sys.argv[1:] or ["scans", "sampler", "store", "ppo", "ppo_wide", "ppo_teacher", "entropy", "popart", "bezier", "f4_nets", "f4_adv", "staged", "f4_cr_adv"]     # name the sets to (re)generate
    for name in todo:
        globals()["gen_" + name]()
