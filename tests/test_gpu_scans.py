"""-m gpu: HIP scans / episode scoring against (a) the reference's golden vectors captured by
tests/golden/make_golden.py and (b) the CPU oracle on larger seeded inputs.

Tolerances: GAE / discounted returns bit-exact (same fp32 op order, no FMA); means use fp64
accumulation rounded once, compared to the reference's fp32 torch means with rtol 2e-6;
max scores exact.
"""
import os

import numpy as np
import pytest
import torch

from oracle import orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("name", ["small", "paired"])
def test_gae_and_discounted_match_reference_bitexact(name):
    from dcd_hip import scans
    g = np.load(os.path.join(G, "scans.npz"))
    r, v, m = g[f"{name}_rewards"], g[f"{name}_value_preds_filled"], g[f"{name}_masks"]
    gamma, lam = g[f"{name}_gamma_lambda"]
    ret = torch.zeros(v.shape, device=DEV)
    scans.gae_returns(dev(r), dev(v), dev(m), ret, float(gamma), float(lam))
    assert np.array_equal(bits(ret.cpu().numpy()[:-1]), bits(g[f"{name}_gae"][:-1]))
    ret.zero_()
    scans.discounted_returns(dev(r), dev(v), dev(m), ret, float(gamma))
    assert np.array_equal(bits(ret.cpu().numpy()), bits(g[f"{name}_disc"]))


@pytest.mark.parametrize("name", ["small", "paired"])
def test_batched_value_loss_matches_reference(name):
    from dcd_hip import scans
    g = np.load(os.path.join(G, "scans.npz"))
    R, v = dev(g[f"{name}_gae"]), dev(g[f"{name}_value_preds_filled"])
    for key, (mode, power, clip) in dict(abs=(0, 1, True), signed=(1, 1, True), pos=(2, 1, True),
                                         abs_p2=(0, 2, True), abs_noclip=(0, 1, False)).items():
        out = scans.batched_value_loss(R, v, mode, power, clip).cpu().numpy()
        np.testing.assert_allclose(out, g[f"{name}_vl_{key}"], rtol=2e-6, atol=1e-7)
        ref = orc.batched_value_loss(g[f"{name}_gae"][..., 0], g[f"{name}_value_preds_filled"][..., 0], mode, power, clip)
        assert np.array_equal(bits(out[:, 0]), bits(ref))


def _big(T=256, B=4096, seed=0):
    rng = np.random.default_rng(seed)
    r = (rng.random((T, B)) * (rng.random((T, B)) < 0.01)).astype(np.float32)
    v = rng.standard_normal((T + 1, B)).astype(np.float32)
    m = (rng.random((T + 1, B)) < 0.99).astype(np.float32)
    m[0] = 1
    m[T] = 0
    return rng, r, v, m


def test_gae_full_size_matches_oracle_bitexact():
    from dcd_hip import scans
    _, r, v, m = _big()
    ret = torch.zeros((257, 4096), device=DEV)
    scans.gae_returns(dev(r), dev(v), dev(m), ret, 0.995, 0.95)
    assert np.array_equal(bits(ret.cpu().numpy()[:-1]), bits(orc.gae_returns(r, v, m, 0.995, 0.95)[:-1]))
    scans.discounted_returns(dev(r), dev(v), dev(m), ret, 0.995)
    assert np.array_equal(bits(ret.cpu().numpy()), bits(orc.discounted_returns(r, v, m, 0.995)))
    # odd sizes: T not a multiple of the unroll, B not a multiple of 64
    r2, v2, m2 = r[:37, :77].copy(), v[:38, :77].copy(), m[:38, :77].copy()
    ret2 = torch.zeros((38, 77), device=DEV)
    scans.gae_returns(dev(r2), dev(v2), dev(m2), ret2, 0.999, 0.9)
    assert np.array_equal(bits(ret2.cpu().numpy()[:-1]), bits(orc.gae_returns(r2, v2, m2, 0.999, 0.9)[:-1]))


STRATS = dict(value_l1=0, positive_value_loss=1, signed_value_loss=2, least_confidence=4, min_margin=5,
              one_step_td_error=6, uniform=7)


@pytest.mark.parametrize("strategy", sorted(STRATS))
def test_episode_scores_match_reference_calls(strategy):
    """Records vs the (actor, seed, score, max, steps, done) sequence the reference's LevelSampler
    passed to _partial_update_seed_score on the same buffers (tests/golden/sampler.npz)."""
    from dcd_hip import scans
    g, s = np.load(os.path.join(G, "scans.npz")), np.load(os.path.join(G, "sampler.npz"))
    d = {k[len("small_"):]: g[k] for k in g.files if k.startswith("small_")}
    T, B = d["rewards"].shape[:2]
    sc = scans.EpisodeScorer(T, B, DEV)
    sc.launch(dev(d["gae"]), dev(d["value_preds"]), dev(d["rewards"]), dev(d["masks"]), dev(d["cliffhanger_masks"]),
              dev(d["level_seeds"]), dev(d["action_log_dist"]), STRATS[strategy], 0.995)
    recs = sc.records()
    keep = recs[(recs["flags"] & 2) == 0]
    calls = s[f"{strategy}_calls"]
    assert len(keep) == len(calls)
    assert np.array_equal(keep["actor"], calls[:, 0]) and np.array_equal(keep["seed"], calls[:, 1])
    assert np.array_equal(keep["num_steps"], calls[:, 4]) and np.array_equal(keep["flags"] & 1, calls[:, 5])
    np.testing.assert_allclose(keep["mean_score"], calls[:, 2], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(keep["max_score"], calls[:, 3], rtol=1e-6, atol=1e-7)


def test_policy_entropy_matches_reference_on_its_defined_lengths():
    """Strategy 3 against the reference's LevelSampler (tests/golden/entropy.npz: every episode has L == 1 or L == A
    steps, the only lengths on which level_sampler.py:283 broadcasts) and against its _average_entropy directly."""
    from dcd_hip import scans
    g = np.load(os.path.join(G, "entropy.npz"))
    d = {k[len("roll_"):]: g[k] for k in g.files if k.startswith("roll_")}
    T, B = d["rewards"].shape[:2]
    sc = scans.EpisodeScorer(T, B, DEV)
    sc.launch(dev(d["value_preds"]), dev(d["value_preds"]), dev(d["rewards"]), dev(d["masks"]), dev(d["cliffhanger_masks"]),
              dev(d["level_seeds"]), dev(d["action_log_dist"]), 3, 0.995)
    recs = sc.records()
    assert set(recs["num_steps"]) == {1, 7}
    keep = recs[(recs["flags"] & 2) == 0]
    calls = g["calls"]
    assert len(keep) == len(calls) > 40
    assert np.array_equal(keep["actor"], calls[:, 0]) and np.array_equal(keep["seed"], calls[:, 1])
    assert np.array_equal(keep["num_steps"], calls[:, 4]) and np.array_equal(keep["flags"] & 1, calls[:, 5])
    np.testing.assert_allclose(keep["mean_score"], calls[:, 2], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(keep["max_score"], calls[:, 3], rtol=1e-5, atol=1e-6)
    for L in (1, 7):
        l = g[f"direct{L}_logits"]
        z = torch.zeros((L + 1, 1), device=DEV)
        m = torch.ones((L + 1, 1), device=DEV)
        m[L] = 0
        one = scans.EpisodeScorer(L, 1, DEV)
        one.launch(z, z, z[:L], m, torch.ones_like(m), torch.ones((L, 1), dtype=torch.int32, device=DEV), dev(l[:, None, :]), 3, 0.995)
        r = one.records()
        assert len(r) == 1 and r["num_steps"][0] == L
        np.testing.assert_allclose([r["mean_score"][0], r["max_score"][0]], g[f"direct{L}_mean_max"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("strategy", [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_episode_scores_full_size_match_oracle(strategy):
    from dcd_hip import scans
    T, B, A = 256, 4096, 7
    rng, r, v, m = _big(T, B, seed=3)
    m = (rng.random((T + 1, B)) < 0.97).astype(np.float32)
    m[0] = 1
    m[T] = 0
    R = (v + 0.3 * rng.standard_normal(v.shape)).astype(np.float32)
    cliff = np.ones_like(m)
    cliff[T][rng.random(B) < 0.5] = 0
    seeds = rng.integers(1, 5000, (T, B)).astype(np.int32)
    ld = rng.standard_normal((T, B, A)).astype(np.float32)
    sc = scans.EpisodeScorer(T, B, DEV)
    sc.launch(dev(R), dev(v), dev(r), dev(m), dev(cliff), dev(seeds), dev(ld), strategy, 0.995)
    recs = sc.records()
    ref = orc.episode_scores(R, v, r, m, cliff, seeds, ld, strategy, 0.995)
    assert len(recs) == len(ref)
    for k in ("actor", "seed", "start_t", "num_steps", "flags"):
        assert np.array_equal(recs[k], ref[k]), k
    if strategy in (3, 4, 5):       # expf/logf differ between device and host libm
        np.testing.assert_allclose(recs["mean_score"], ref["mean_score"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(recs["max_score"], ref["max_score"], rtol=1e-5, atol=1e-6)
    else:
        assert np.array_equal(bits(recs["mean_score"]), bits(ref["mean_score"]))
        assert np.array_equal(bits(recs["max_score"]), bits(ref["max_score"]))
    assert np.array_equal(bits(recs["reward_sum"]), bits(ref["reward_sum"]))
    # grounded second pass
    n = len(recs)
    gv = rng.standard_normal(n).astype(np.float32)
    for pos, dense in ((0, 0), (1, 0), (1, 1)):
        mean, mx = sc.grounded(n, dev(gv), dev(v), pos, dense)
        rm, rx = orc.grounded_scores(ref, gv, v, pos, dense)
        assert np.array_equal(bits(mean.cpu().numpy()), bits(rm)) and np.array_equal(bits(mx.cpu().numpy()), bits(rx))


@pytest.mark.parametrize("T,B", [(1, 32), (2, 64), (37, 32), (256, 96), (300, 64), (513, 160), (40, 77)])
@pytest.mark.parametrize("strategy", [1, 6, 8])
def test_episode_scores_staged_path_shapes(strategy, T, B):
    """The LDS-staged scoring path (B a multiple of 32) over one, two and three time tiles, partial tiles, blocks that
    span several count workgroups, and (B = 77) the one-env-per-lane path it falls back to; a done on the last step or not."""
    from dcd_hip import scans
    rng = np.random.default_rng(100 * T + B + strategy)
    r = (rng.random((T, B)) * (rng.random((T, B)) < 0.05)).astype(np.float32)
    v = rng.standard_normal((T + 1, B)).astype(np.float32)
    m = (rng.random((T + 1, B)) < 0.9).astype(np.float32)
    m[T][rng.random(B) < 0.5] = 0
    R = (v + 0.3 * rng.standard_normal(v.shape)).astype(np.float32)
    cliff = (rng.random((T + 1, B)) < 0.7).astype(np.float32)
    seeds = rng.integers(1, 5000, (T, B)).astype(np.int32)
    sc = scans.EpisodeScorer(T, B, DEV)
    sc.launch(dev(R), dev(v), dev(r), dev(m), dev(cliff), dev(seeds), None, strategy, 0.995)
    recs = sc.records()
    ref = orc.episode_scores(R, v, r, m, cliff, seeds, None, strategy, 0.995)
    assert len(recs) == len(ref)
    for k in ("actor", "seed", "start_t", "num_steps", "flags"):
        assert np.array_equal(recs[k], ref[k]), k
    for k in ("mean_score", "max_score", "reward_sum"):
        assert np.array_equal(bits(recs[k]), bits(ref[k])), k


def test_return_stats_and_regret_match_oracle():
    from dcd_hip import scans
    rng = np.random.default_rng(9)
    T, B = 256, 1000
    info = ((rng.random((T, B)) < 0.02) * 1 + (rng.random((T, B)) < 0.1) * 2).astype(np.uint8)
    ep = (rng.random((T, B)) * (rng.random((T, B)) < 0.5)).astype(np.float32)
    mean, mx, n = scans.rollout_return_stats(dev(ep), dev(info))
    rm, rx, rn = orc.rollout_return_stats(ep, info)
    assert np.array_equal(bits(mean.cpu().numpy()[:, 0]), bits(rm)) and np.array_equal(bits(mx.cpu().numpy()[:, 0]), bits(rx))
    assert np.array_equal(n.cpu().numpy(), rn)
    a = [rng.random(B).astype(np.float32) for _ in range(4)]
    for mode in (0, 1, 2):
        out = scans.env_return(*[dev(x) for x in a], mode=mode).cpu().numpy()
        assert np.array_equal(bits(out), bits(orc.env_return(*a, mode=mode)))


def test_batched_scans_equal_the_single_launches():
    """dcd_gae_returns_batch / dcd_episode_scores_batch / dcd_rollout_return_stats_batch over two rollouts (PAIRED's two
    students) == the single-rollout entry points, bit for bit; staged scoring path and the one-env-per-lane fallback."""
    from dcd_hip import scans
    for (T, B, strategy, with_logits) in ((256, 4096, 1, False), (37, 96, 6, False), (40, 77, 1, False), (64, 64, 4, True)):
        sets = []
        for k in range(2):
            rng, r, v, m = _big(T, B, seed=20 + k)
            m = (rng.random((T + 1, B)) < 0.97).astype(np.float32)
            m[0] = 1
            m[T][rng.random(B) < 0.6] = 0
            cliff = (rng.random((T + 1, B)) < 0.8).astype(np.float32)
            seeds = rng.integers(1, 999, (T, B)).astype(np.int32)
            info = ((rng.random((T, B)) < 0.03) * 1).astype(np.uint8)
            ep = rng.random((T, B)).astype(np.float32)
            ld = rng.standard_normal((T, B, 7)).astype(np.float32) if with_logits else None
            sets.append(dict(r=dev(r), v=dev(v), m=dev(m), cliff=dev(cliff), seeds=dev(seeds), info=dev(info), ep=dev(ep),
                             ld=dev(ld) if ld is not None else None))
        single, batch = [], []
        for s_ in sets:                                               # single launches
            ret = torch.zeros((T + 1, B), device=DEV)
            scans.gae_returns(s_["r"], s_["v"], s_["m"], ret, 0.995, 0.95)
            sc = scans.EpisodeScorer(T, B, DEV)
            sc.launch(ret, s_["v"], s_["r"], s_["m"], s_["cliff"], s_["seeds"], s_["ld"], strategy, 0.995)
            st = scans.rollout_return_stats(s_["ep"], s_["info"])
            single.append((ret, sc.records(), [x.clone() for x in st]))
        rets = [torch.zeros((T + 1, B), device=DEV) for _ in sets]
        scans.gae_returns_batch([(s_["r"], s_["v"], s_["m"], rets[k]) for k, s_ in enumerate(sets)], 0.995, 0.95)
        scs = [scans.EpisodeScorer(T, B, DEV) for _ in sets]
        scans.EpisodeScorer.launch_batch(scs, [(rets[k], s_["v"], s_["r"], s_["m"], s_["cliff"], s_["seeds"], s_["ld"]) for k, s_ in enumerate(sets)], strategy, 0.995)
        outs = [(torch.zeros((B, 1), device=DEV), torch.zeros((B, 1), device=DEV), torch.zeros(B, dtype=torch.int32, device=DEV)) for _ in sets]
        scans.rollout_return_stats_batch([(s_["ep"], s_["info"]) + outs[k] for k, s_ in enumerate(sets)])
        for k in range(2):
            assert torch.equal(rets[k][:-1], single[k][0][:-1])
            a, b = scs[k].records(), single[k][1]
            assert len(a) == len(b) > 0 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
            for x, y in zip(outs[k], single[k][2]):
                assert torch.equal(x.view(-1), y.view(-1))


@pytest.mark.parametrize("T,B,with_scores", [(256, 4096, True), (64, 96, True), (33, 100, True), (64, 256, False), (300, 64, True), (129, 32, False)])
def test_rollout_tail_batch_equals_the_separate_calls(T, B, with_scores):
    """dcd_rollout_tail_batch ({GAE, return statistics, segment counts} of two rollouts side by side in one launch, then the
    scoring) == gae_returns_batch + rollout_return_stats_batch + EpisodeScorer.launch_batch, bit for bit — returns, statistics,
    records and record counts; B = 100 is not a multiple of 32: the scoring falls back to the one-env-per-lane path and the
    counts are not fused; with_scores = False: the single-launch form without level replay."""
    from dcd_hip import SCORE_STRATEGY, scans
    g = torch.Generator().manual_seed(T * 7 + B)
    def rollout():
        r = dict(rewards=(torch.rand(T, B, generator=g) * (torch.rand(T, B, generator=g) < 0.05)).to(DEV),
                 value_preds=torch.randn(T + 1, B, generator=g).to(DEV),
                 masks=(torch.rand(T + 1, B, generator=g) > 0.03).float().to(DEV), cliff=(torch.rand(T + 1, B, generator=g) > 0.01).float().to(DEV),
                 seeds=torch.randint(0, 1000, (T, B), generator=g).to(torch.int32).to(DEV),
                 ep_ret=torch.rand(T, B, generator=g).to(DEV), info=((torch.rand(T, B, generator=g) < 0.03) * 1).to(torch.uint8).to(DEV))
        r["masks"][0] = 1.0
        return r
    rolls = [rollout(), rollout()]
    def run(fused):
        outs = []
        for r in rolls:
            outs.append(dict(returns=torch.zeros(T + 1, B, device=DEV), mean=torch.zeros(B, 1, device=DEV), mx=torch.zeros(B, 1, device=DEV),
                             n=torch.zeros(B, dtype=torch.int32, device=DEV), scorer=scans.EpisodeScorer(T, B, DEV)))
        gae = [(r["rewards"], r["value_preds"], r["masks"], o["returns"]) for r, o in zip(rolls, outs)]
        st = [(r["ep_ret"], r["info"], o["mean"], o["mx"], o["n"]) for r, o in zip(rolls, outs)]
        sa = [(o["returns"], r["value_preds"], r["rewards"], r["masks"], r["cliff"], r["seeds"], None) for r, o in zip(rolls, outs)]
        sc = [o["scorer"] for o in outs]
        strat = SCORE_STRATEGY["positive_value_loss"]
        if fused:
            scans.rollout_tail_batch(gae, st, 0.995, 0.95, scorers=sc if with_scores else None, score_argsets=sa if with_scores else None, strategy=strat)
        else:
            scans.gae_returns_batch(gae, 0.995, 0.95)
            scans.rollout_return_stats_batch(st)
            if with_scores:
                scans.EpisodeScorer.launch_batch(sc, sa, strat, 0.995)
        torch.cuda.synchronize()
        return outs
    a, b = run(True), run(False)
    for x, y in zip(a, b):
        for k in ("returns", "mean", "mx", "n"):
            assert torch.equal(x[k], y[k]), k
        if with_scores:
            rx, ry = x["scorer"].records(), y["scorer"].records()
            assert len(rx) == len(ry) > 0 and rx.tobytes() == ry.tobytes()
