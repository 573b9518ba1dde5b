"""-m gpu: the BENCHMARKED scan kernels — `rollout_tail_kernel` (GAE + return statistics + segment counts of both PAIRED students in
one launch) and `seg_score_lds_kernel<ST>` (the LDS-staged episode scoring, B a multiple of 32) — against outputs of the reference
itself: tests/golden/staged.npz holds rollouts at T = 256 / B = 64 and T = 300 / B = 96 (one and two time tiles, two and three
32-env blocks), their returns through the imported RolloutStorage.compute_returns (storage.py:233-256) and the (actor, seed, score,
max, steps, done) sequence the imported LevelSampler passed to _partial_update_seed_score (level_sampler.py:486-578).

Tolerances as in test_gpu_scans.py: returns bit-exact; mean scores rtol 2e-6 (fp64 accumulation rounded once vs torch's fp32
vectorised mean); max scores exact up to the fp64 -> fp32 rounding of the golden's `.item()`.
"""
import os
import types

import numpy as np
import pytest
import torch

from oracle import orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
STRATS = dict(positive_value_loss=1, value_l1=0, signed_value_loss=2, one_step_td_error=6)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _load(name):
    g = np.load(os.path.join(G, "staged.npz"))
    d = {k: g[f"{name}_{k}"] for k in ("rewards", "value_preds", "masks", "cliffhanger_masks", "level_seeds", "gae")}
    return g, d


def _check_calls(recs, g, name, strategy):
    keep = recs[(recs["flags"] & 2) == 0]
    ci, cf = g[f"{name}_{strategy}_calls_i"], g[f"{name}_{strategy}_calls_f"]
    assert len(keep) == len(ci) > 400
    assert np.array_equal(keep["actor"], ci[:, 0]) and np.array_equal(keep["seed"], ci[:, 1])
    assert np.array_equal(keep["num_steps"], ci[:, 2]) and np.array_equal(keep["flags"] & 1, ci[:, 3])
    np.testing.assert_allclose(keep["mean_score"], cf[:, 0], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(keep["max_score"], cf[:, 1], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("shape", ["a", "b"])
@pytest.mark.parametrize("strategy", sorted(STRATS))
def test_rollout_tail_batch_matches_reference(shape, strategy):
    """dcd_rollout_tail_batch with the two golden rollouts of one shape as PAIRED's two students: rollout_tail_kernel's returns
    bit-exact against compute_returns, seg_score_lds_kernel's records against the reference's call sequence; the return statistics
    riding in the same launch against the oracle (the runner that computes them is not importable)."""
    from dcd_hip import scans
    names = [shape + "0", shape + "1"]
    loaded = [_load(n) for n in names]
    g = loaded[0][0]
    gamma, lam = (float(x) for x in g["gamma_lambda"])
    T, B = loaded[0][1]["rewards"].shape
    assert B % 32 == 0                                  # the staged path
    rng = np.random.default_rng(T + B)
    rolls, outs = [], []
    for _, d in loaded:
        info = ((rng.random((T, B)) < 0.03) * 1).astype(np.uint8)
        ep = rng.random((T, B)).astype(np.float32)
        rolls.append(dict(r=dev(d["rewards"]), v=dev(d["value_preds"]), m=dev(d["masks"]), c=dev(d["cliffhanger_masks"]),
                          s=dev(d["level_seeds"]), ep=dev(ep), info=dev(info), ep_np=ep, info_np=info))
        outs.append(dict(ret=torch.zeros(T + 1, B, device=DEV), mean=torch.zeros(B, 1, device=DEV), mx=torch.zeros(B, 1, device=DEV),
                         n=torch.zeros(B, dtype=torch.int32, device=DEV), sc=scans.EpisodeScorer(T, B, DEV)))
    scans.rollout_tail_batch([(r["r"], r["v"], r["m"], o["ret"]) for r, o in zip(rolls, outs)],
                             [(r["ep"], r["info"], o["mean"], o["mx"], o["n"]) for r, o in zip(rolls, outs)], gamma, lam,
                             scorers=[o["sc"] for o in outs],
                             score_argsets=[(o["ret"], r["v"], r["r"], r["m"], r["c"], r["s"], None) for r, o in zip(rolls, outs)],
                             strategy=STRATS[strategy])
    for name, (_, d), r, o in zip(names, loaded, rolls, outs):
        assert np.array_equal(bits(o["ret"].cpu().numpy()[:-1]), bits(d["gae"][:-1])), name
        _check_calls(o["sc"].records(), g, name, strategy)
        rm, rx, rn = orc.rollout_return_stats(r["ep_np"], r["info_np"])
        assert np.array_equal(bits(o["mean"].cpu().numpy()[:, 0]), bits(rm)) and np.array_equal(bits(o["mx"].cpu().numpy()[:, 0]), bits(rx))
        assert np.array_equal(o["n"].cpu().numpy(), rn)


@pytest.mark.parametrize("name", ["a0", "a1", "b0", "b1"])
@pytest.mark.parametrize("strategy", sorted(STRATS))
def test_episode_scorer_staged_path_matches_reference(name, strategy):
    """EpisodeScorer.launch (seg_count_par_kernel + seg_score_lds_kernel) on the reference's own returns."""
    from dcd_hip import scans
    g, d = _load(name)
    T, B = d["rewards"].shape
    sc = scans.EpisodeScorer(T, B, DEV)
    sc.launch(dev(d["gae"]), dev(d["value_preds"]), dev(d["rewards"]), dev(d["masks"]), dev(d["cliffhanger_masks"]),
              dev(d["level_seeds"]), None, STRATS[strategy], float(g["gamma_lambda"][0]))
    _check_calls(sc.records(), g, name, strategy)


@pytest.mark.parametrize("name", ["a0", "b1"])
@pytest.mark.parametrize("strategy", ["grounded_positive_value_loss", "grounded_signed_value_loss", "positive_value_loss"])
def test_level_sampler_on_device_rollouts_matches_reference_state(name, strategy):
    """The product's LevelSampler.update_with_rollouts on device tensors (seg_score_lds_kernel<8> for the grounded strategies:
    records with reward sums, the grounded advantage on the host) against the sampler state the reference reached."""
    from level_replay import LevelSampler
    g, d = _load(name)
    T, B = d["rewards"].shape

    class _Space:
        n = 7

    ro = types.SimpleNamespace(device=torch.device(DEV), returns=dev(d["gae"])[..., None], value_preds=dev(d["value_preds"])[..., None],
                               rewards=dev(d["rewards"])[..., None], masks=dev(d["masks"])[..., None],
                               cliffhanger_masks=dev(d["cliffhanger_masks"])[..., None], level_seeds=dev(d["level_seeds"])[..., None],
                               action_log_dist=torch.zeros(T, B, 7, device=DEV), use_popart=False)
    ls = LevelSampler(list(range(1, 12)), None, _Space(), num_actors=B, strategy=strategy, score_transform="rank", temperature=0.1,
                      staleness_coef=0.3, rho=0.5, replay_prob=0.5, alpha=1.0, gamma=float(g["gamma_lambda"][0]))
    ls.update_with_rollouts(ro)
    np.testing.assert_allclose(ls.seed_scores, g[f"{name}_{strategy}_seed_scores"], rtol=3e-6, atol=3e-7)
    if ls.grounded_values is not None:
        np.testing.assert_allclose(ls.grounded_values, g[f"{name}_{strategy}_grounded"], rtol=3e-6, atol=3e-7)
