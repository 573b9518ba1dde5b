"""CPU, world_size 2 over gloo: the exchange steps of the sharded path.

  * PPO with gradient all-reduce + global advantage moments on 2 ranks x B/2 envs == one process
    with B envs (same parameters after the update, fp32 tolerance);
  * episode-score records all-gathered to every rank in global actor order; rank 0's LevelSampler
    fed with them == a single-process sampler on the concatenated batch.
"""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = os.path.join(HERE, "golden")
T, B = 12, 4


def _setup(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, "dcd-isaac_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _eight_envs(g):
    """The golden 4-env rollout widened to 8 distinct envs: columns 4..7 are columns 3,0,1,2 with other returns / values."""
    out = {}
    for k in g.files:
        a = g[k]
        if k in ("img", "direction", "masks", "rewards", "hidden", "act_actions", "act_values", "action_log_probs", "returns"):
            b = np.roll(a, 1, axis=1)
            if k in ("returns", "act_values", "rewards"):
                b = (b * 0.5 + 0.1).astype(a.dtype)
            a = np.concatenate([a, b], axis=1)
        elif k == "next_value":
            a = np.concatenate([a, np.roll(a, 1, axis=0) * 0.5], axis=0)
        out[k] = a

    class _G(dict):
        files = list(out)
    return _G(out)


def _build(g, envs, process_group):
    from algos import ACAgent, PPO, RolloutStorage
    from envs import spaces
    from models import MultigridNetwork
    obs = {"image": spaces.Box(0, 255, (3, 5, 5), "uint8"), "direction": spaces.Box(0, 3, (1,), "uint8")}
    net = MultigridNetwork(obs, spaces.Discrete(7), scalar_fc=5, scalar_dim=4, recurrent_arch="lstm", recurrent_hidden_size=256)
    net.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w_") and not k.startswith("w_after_")})
    n = len(envs)
    st = RolloutStorage(net, T, n, obs, spaces.Discrete(7), 256, recurrent_arch="lstm", use_proper_time_limits=False)
    sel = lambda a: torch.from_numpy(np.ascontiguousarray(a[:, envs]))
    st.obs["image"].copy_(sel(g["img"])); st.obs["direction"].copy_(sel(g["direction"]))
    st.masks.copy_(sel(g["masks"])); st.rewards.copy_(sel(g["rewards"]))
    st.recurrent_hidden_states.copy_(sel(g["hidden"])); st.actions.copy_(sel(g["act_actions"]))
    st.value_preds[:-1].copy_(sel(g["act_values"])); st.value_preds[-1].copy_(torch.from_numpy(g["next_value"][envs]))
    st.action_log_probs.copy_(sel(g["action_log_probs"])); st.returns.copy_(sel(g["returns"]))
    algo = PPO(net, clip_param=0.2, ppo_epoch=2, num_mini_batch=1, value_loss_coef=0.5, entropy_coef=0.01, lr=1e-4,
               eps=1e-5, max_grad_norm=0.5, clip_value_loss=True, process_group=process_group)
    return ACAgent(algo, st), net


def _ppo_worker(rank, world, port, out_path):
    _setup(rank, world, port)
    g = np.load(os.path.join(G, "ppo.npz"))
    envs = [0, 1] if rank == 0 else [2, 3]
    agent, net = _build(g, envs, dist.group.WORLD)
    torch.manual_seed(2)
    vl, al, ent, _ = agent.update()
    if rank == 0:
        torch.save({"sd": net.state_dict(), "losses": (vl, al, ent)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def _ppo8_worker(rank, world, port, out_path):
    _setup(rank, world, port)
    from algos import PPO
    PPO.BUCKET_BYTES = 256 << 10                                   # the 1.7 MB student gradient in 7 buckets
    g = _eight_envs(np.load(os.path.join(G, "ppo.npz")))
    agent, net = _build(g, [rank], dist.group.WORLD)
    torch.manual_seed(2)
    vl, al, ent, _ = agent.update()
    assert len(agent.algo._bucket_bounds) >= 6 and all(agent.algo._launched)
    if rank == 0:
        torch.save({"sd": net.state_dict(), "losses": (vl, al, ent)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_ppo_eight_ranks_bucketed_exchange_equals_single_process():
    """8 ranks x 1 env, gradients in a flat buffer exchanged as 7 asynchronous bucket all-reduces launched from the
    backward hooks, loss statistics in the last bucket's tail == one process with the 8 envs."""
    for p in (ROOT, os.path.join(ROOT, "dcd-isaac_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    g = _eight_envs(np.load(os.path.join(G, "ppo.npz")))
    agent, net = _build(g, list(range(8)), None)
    torch.manual_seed(2)
    ref_losses = agent.update()[:3]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r0.pt")
        mp.spawn(_ppo8_worker, args=(8, 29543, out), nprocs=8, join=True)
        got = torch.load(out, weights_only=False)
    np.testing.assert_allclose(got["losses"], ref_losses, rtol=1e-4, atol=1e-7)
    for k, v in net.state_dict().items():
        np.testing.assert_allclose(got["sd"][k].numpy(), v.numpy(), rtol=0, atol=3e-6, err_msg=k)


def test_sharded_ppo_equals_single_process():
    for p in (ROOT, os.path.join(ROOT, "dcd-isaac_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    g = np.load(os.path.join(G, "ppo.npz"))
    agent, net = _build(g, [0, 1, 2, 3], None)
    torch.manual_seed(2)
    ref_losses = agent.update()[:3]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r0.pt")
        mp.spawn(_ppo_worker, args=(2, 29533, out), nprocs=2, join=True)
        got = torch.load(out, weights_only=False)
    np.testing.assert_allclose(got["losses"], ref_losses, rtol=1e-4, atol=1e-7)
    for k, v in net.state_dict().items():
        np.testing.assert_allclose(got["sd"][k].numpy(), v.numpy(), rtol=0, atol=3e-6, err_msg=k)


def _records_worker(rank, world, port, out_path):
    _setup(rank, world, port)
    from oracle import orc
    from util import distributed as D
    g = np.load(os.path.join(G, "scans.npz"))
    d = {k[len("small_"):]: g[k][..., 0] if g[k].ndim == 3 and g[k].shape[-1] == 1 else g[k] for k in g.files if k.startswith("small_")}
    half = slice(0, 4) if rank == 0 else slice(4, 8)
    recs = orc.episode_scores(d["gae"][:, half], d["value_preds"][:, half], d["rewards"][:, half], d["masks"][:, half],
                              d["cliffhanger_masks"][:, half], d["level_seeds"][:, half], None, 1, 0.995)
    parts = D.all_gather_records(recs, "cpu")
    regret = D.all_gather_tensor(torch.full((4, 1), float(rank)))
    assert regret.shape == (8, 1) and regret[:4].eq(0).all() and regret[4:].eq(1).all()
    if rank == 0:
        np.save(out_path, np.array([len(p) for p in parts]))
        np.save(out_path + ".recs.npy", np.concatenate([np.ascontiguousarray(p).view(np.int32).reshape(-1, 8) for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


class _Space:
    n = 7


def test_gathered_records_feed_one_sampler():
    from level_replay import LevelSampler
    from oracle import orc
    g = np.load(os.path.join(G, "scans.npz"))
    sq = lambda a: a[..., 0]
    d = {k[len("small_"):]: g[k] for k in g.files if k.startswith("small_")}
    whole = orc.episode_scores(sq(d["gae"]), sq(d["value_preds"]), sq(d["rewards"]), sq(d["masks"]),
                               sq(d["cliffhanger_masks"]), sq(d["level_seeds"]), None, 1, 0.995)
    with tempfile.TemporaryDirectory() as dd:
        out = os.path.join(dd, "counts.npy")
        mp.spawn(_records_worker, args=(2, 29534, out), nprocs=2, join=True)
        counts = np.load(out)
        flat = np.load(out + ".recs.npy")
    parts, off = [], 0
    for c in counts:
        parts.append(flat[off:off + c].copy().view(orc.EPISODE_REC).reshape(c))
        off += c
    assert sum(counts) == len(whole)
    mk = lambda: LevelSampler(list(range(1, 12)), None, _Space(), num_actors=8, strategy="positive_value_loss",
                              score_transform="rank", temperature=0.1, staleness_coef=0.3, alpha=1.0)
    a, b = mk(), mk()
    a.update_with_records(whole)
    for r, part in enumerate(parts):
        b.update_with_records(part, actor_offset=4 * r)            # global actor = rank * B + local actor
    assert np.array_equal(a.seed_scores, b.seed_scores) and np.array_equal(a.partial_seed_scores, b.partial_seed_scores)
    assert np.array_equal(a.partial_seed_steps, b.partial_seed_steps)


def _paired_regret(mean, mx, adv_mean, adv_max, mode=0):
    """Host stand-in for the HIP env-return kernel (PAIRED: max antagonist return - mean protagonist return, >= 0)."""
    return (adv_max - mean).clamp(min=0).unsqueeze(-1)


def _teacher_return(runner_cls, info, adv_info, world, group, updates):
    import types
    from dcd_hip import scans
    from envs.runners.adversarial_runner import RunningMeanStd
    args = types.SimpleNamespace(ued_algo="paired", adv_normalize_returns=True, adv_clip_reward=None)
    me = types.SimpleNamespace(args=args, is_paired=True, world_size=world, distributed=group is not None, process_group=group, env_return_rms=RunningMeanStd(shape=()))
    keep, scans.env_return = scans.env_return, _paired_regret        # collective logic only: no GPU in this test
    try:
        for u in range(updates):
            out = runner_cls._compute_env_return(me, {k: v + u for k, v in info.items()}, {k: v * (1 + u) for k, v in adv_info.items()})
    finally:
        scans.env_return = keep
    return out, me.env_return_rms


def _infos(envs):
    gen = torch.Generator().manual_seed(7)
    full = {k: torch.rand(8, generator=gen) for k in ("mean_return", "max_return", "adv_mean", "adv_max")}
    return ({"mean_return": full["mean_return"][envs], "max_return": full["max_return"][envs]},
            {"mean_return": full["adv_mean"][envs], "max_return": full["adv_max"][envs] + 0.5})


def _rms_worker(rank, world, port, out_path):
    _setup(rank, world, port)
    from envs.runners.adversarial_runner import AdversarialRunner
    envs = list(range(4)) if rank == 0 else list(range(4, 8))
    ret, rms = _teacher_return(AdversarialRunner, *_infos(envs), world, dist.group.WORLD, updates=3)
    parts = [torch.empty_like(ret) for _ in range(world)]
    dist.all_gather(parts, ret)
    if rank == 0:
        torch.save({"ret": torch.cat(parts), "mean": float(rms.mean), "var": float(rms.var), "count": float(rms.count)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_adv_normalize_returns_uses_job_wide_moments():
    """the teacher-return RMS (adversarial_runner.py:666-668) over 2 ranks x 4 envs == one process, 8 envs."""
    for p in (ROOT, os.path.join(ROOT, "dcd-isaac_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from envs.runners.adversarial_runner import AdversarialRunner
    want, rms = _teacher_return(AdversarialRunner, *_infos(list(range(8))), 1, None, updates=3)
    with tempfile.TemporaryDirectory() as dd:
        out = os.path.join(dd, "rms.pt")
        mp.spawn(_rms_worker, args=(2, 29537, out), nprocs=2, join=True)
        got = torch.load(out, weights_only=True)
    assert got["mean"] == float(rms.mean) and got["var"] == float(rms.var) and got["count"] == float(rms.count)
    assert torch.equal(got["ret"], want)


def _one_rank_worker(rank, world, port, out_path):
    _setup(rank, world, port)
    g = np.load(os.path.join(G, "ppo.npz"))
    agent, net = _build(g, [0, 1, 2, 3], dist.group.WORLD)
    assert agent.algo.distributed and agent.algo.world_size == 1
    torch.manual_seed(2)
    vl, al, ent, _ = agent.update()
    assert all(agent.algo._launched)                               # the bucket all-reduces ran (as the identity)
    torch.save({"sd": net.state_dict(), "losses": (vl, al, ent)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_group_takes_the_exchange_path_and_equals_no_group_bit_for_bit():
    """A process group of ONE rank runs every exchange step (weight broadcast, hook-launched bucket all-reduces, advantage
    moments, statistics tail) — each the identity — and must therefore equal the run without a group exactly: what
    tests/test_gpu_rccl_single.py relies on to exercise RCCL on a one-GPU box."""
    for p in (ROOT, os.path.join(ROOT, "dcd-isaac_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    g = np.load(os.path.join(G, "ppo.npz"))
    agent, net = _build(g, [0, 1, 2, 3], None)
    torch.manual_seed(2)
    ref_losses = agent.update()[:3]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r0.pt")
        mp.spawn(_one_rank_worker, args=(1, 29547, out), nprocs=1, join=True)
        got = torch.load(out, weights_only=False)
    assert tuple(got["losses"]) == tuple(ref_losses)
    for k, v in net.state_dict().items():
        assert torch.equal(got["sd"][k], v), k


def _popart_worker(rank, world, port, out_path):
    _setup(rank, world, port)
    from models.popart import PopArt
    torch.manual_seed(0)
    pa = PopArt(8, 1)
    x = torch.randn(64, 1, generator=torch.Generator().manual_seed(3)) * 3 + 1
    for it in range(3):
        pa.update(x[rank * 32:(rank + 1) * 32] * (1 + it), group=dist.group.WORLD)
    torch.save({k: v.clone() for k, v in pa.state_dict().items()}, out_path + str(rank))
    dist.barrier()
    dist.destroy_process_group()


def test_popart_moments_are_job_wide():
    """PPO under a process group moves PopArt by the moments of the job-wide minibatch (models/popart.py::update(group=)): two
    ranks x 32 returns end in the SAME state (bit for bit) == one process with the 64 returns (fp32 rounding of the two
    half-means)."""
    for p in (ROOT, os.path.join(ROOT, "dcd-isaac_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from models.popart import PopArt
    torch.manual_seed(0)
    pa = PopArt(8, 1)
    x = torch.randn(64, 1, generator=torch.Generator().manual_seed(3)) * 3 + 1
    for it in range(3):
        pa.update(x * (1 + it))
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "pa")
        mp.spawn(_popart_worker, args=(2, 29549, out), nprocs=2, join=True)
        a, b = torch.load(out + "0"), torch.load(out + "1")
    for k, v in pa.state_dict().items():
        assert torch.equal(a[k], b[k]), k
        np.testing.assert_allclose(a[k].numpy(), v.numpy(), rtol=2e-6, atol=1e-9, err_msg=k)


def test_discarded_update_without_backward_equals_the_one_with():
    """PPO.update(discard_grad=True) skips the backward pass unless gradient norms are logged: same losses, same (untouched)
    weights, same position of torch's generator as the update that computes and discards the gradients (ppo.py:129-130)."""
    for p in (ROOT, os.path.join(ROOT, "dcd-isaac_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    g = np.load(os.path.join(G, "ppo.npz"))
    out = []
    for log_norm in (True, False):
        agent, net = _build(g, [0, 1, 2, 3], None)
        agent.algo.log_grad_norm = log_norm
        before = {k: v.clone() for k, v in net.state_dict().items()}
        torch.manual_seed(2)
        vl, al, ent, info = agent.algo.update(agent.storage, discard_grad=True)
        out.append((vl, al, ent, torch.get_rng_state().clone()))
        for k, v in net.state_dict().items():
            assert torch.equal(v, before[k]), k
        assert ("grad_norms" in info) == log_norm
    assert out[0][:3] == out[1][:3] and torch.equal(out[0][3], out[1][3])
