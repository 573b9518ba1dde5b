"""CPU: pins oracle/scan_oracle.c to the reference's own outputs (tests/golden/*.npz, produced by
importing the reference's algos/ and level_replay/ in the build container)."""
import os

import numpy as np
import pytest

from oracle import orc

G = os.path.join(os.path.dirname(__file__), "golden")
sq = lambda a: a[..., 0]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("name", ["small", "paired"])
def test_gae_discounted_bitexact_vs_reference(name):
    g = np.load(os.path.join(G, "scans.npz"))
    r, v, m = sq(g[f"{name}_rewards"]), sq(g[f"{name}_value_preds_filled"]), sq(g[f"{name}_masks"])
    gamma, lam = g[f"{name}_gamma_lambda"]
    assert np.array_equal(bits(orc.gae_returns(r, v, m, gamma, lam)[:-1]), bits(sq(g[f"{name}_gae"])[:-1]))
    assert np.array_equal(bits(orc.discounted_returns(r, v, m, gamma)), bits(sq(g[f"{name}_disc"])))


@pytest.mark.parametrize("name", ["small", "paired"])
def test_value_loss_vs_reference(name):
    g = np.load(os.path.join(G, "scans.npz"))
    R, v = sq(g[f"{name}_gae"]), sq(g[f"{name}_value_preds_filled"])
    for key, (mode, power, clip) in dict(abs=(0, 1, True), signed=(1, 1, True), pos=(2, 1, True),
                                         abs_p2=(0, 2, True), abs_noclip=(0, 1, False)).items():
        np.testing.assert_allclose(orc.batched_value_loss(R, v, mode, power, clip), sq(g[f"{name}_vl_{key}"]),
                                   rtol=2e-6, atol=1e-7)


STRATS = dict(value_l1=0, positive_value_loss=1, signed_value_loss=2, gae=2, least_confidence=4, min_margin=5,
              one_step_td_error=6, uniform=7)


@pytest.mark.parametrize("strategy", sorted(STRATS))
def test_episode_records_vs_reference_calls(strategy):
    g, s = np.load(os.path.join(G, "scans.npz")), np.load(os.path.join(G, "sampler.npz"))
    d = {k[len("small_"):]: g[k] for k in g.files if k.startswith("small_")}
    recs = orc.episode_scores(sq(d["gae"]), sq(d["value_preds"]), sq(d["rewards"]), sq(d["masks"]),
                              sq(d["cliffhanger_masks"]), sq(d["level_seeds"]), d["action_log_dist"],
                              STRATS[strategy], 0.995)
    keep = recs[(recs["flags"] & 2) == 0]
    calls = s[f"{strategy}_calls"]
    assert len(keep) == len(calls) and len(recs) > len(keep) > 20
    assert np.array_equal(keep["actor"], calls[:, 0]) and np.array_equal(keep["seed"], calls[:, 1])
    assert np.array_equal(keep["num_steps"], calls[:, 4]) and np.array_equal(keep["flags"] & 1, calls[:, 5])
    np.testing.assert_allclose(keep["mean_score"], calls[:, 2], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(keep["max_score"], calls[:, 3], rtol=1e-6, atol=1e-7)


def test_policy_entropy_vs_reference_on_its_defined_lengths():
    """level_sampler.py:283 broadcasts [L,A] against [L]: defined for L in {1, A} only.  tests/golden/entropy.npz holds a
    rollout whose every episode has one of those lengths, scored by the reference's LevelSampler, and direct calls of
    _average_entropy; exp/log-based scores: rtol 1e-5 (host libm vs torch's vectorised exp)."""
    g = np.load(os.path.join(G, "entropy.npz"))
    d = {k[len("roll_"):]: g[k] for k in g.files if k.startswith("roll_")}
    recs = orc.episode_scores(sq(d["value_preds"]), sq(d["value_preds"]), sq(d["rewards"]), sq(d["masks"]),
                              sq(d["cliffhanger_masks"]), sq(d["level_seeds"]), d["action_log_dist"], 3, 0.995)
    assert set(recs["num_steps"]) == {1, 7}
    keep = recs[(recs["flags"] & 2) == 0]
    calls = g["calls"]
    assert len(keep) == len(calls) > 40
    assert np.array_equal(keep["actor"], calls[:, 0]) and np.array_equal(keep["seed"], calls[:, 1])
    assert np.array_equal(keep["num_steps"], calls[:, 4]) and np.array_equal(keep["flags"] & 1, calls[:, 5])
    np.testing.assert_allclose(keep["mean_score"], calls[:, 2], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(keep["max_score"], calls[:, 3], rtol=1e-5, atol=1e-6)
    for L in (1, 7):                                   # one episode on its own
        l = g[f"direct{L}_logits"]
        z = np.zeros((L + 1, 1), np.float32)
        m = np.ones((L + 1, 1), np.float32)
        m[L] = 0
        r = orc.episode_scores(z, z, z[:L], m, np.ones_like(m), np.ones((L, 1), np.int32), l[:, None, :], 3)
        assert len(r) == 1 and r["num_steps"][0] == L
        np.testing.assert_allclose([r["mean_score"][0], r["max_score"][0]], g[f"direct{L}_mean_max"], rtol=1e-5, atol=1e-6)


def test_policy_entropy_other_lengths_is_the_per_step_entropy():
    """Lengths the reference's expression rejects: mean_t / max_t of sum_a(-p log p) / log A."""
    rng = np.random.default_rng(0)
    L, A = 5, 7
    l = rng.standard_normal((L, 1, A)).astype(np.float32)
    z = np.zeros((L + 1, 1), np.float32)
    m = np.ones((L + 1, 1), np.float32)
    m[L] = 0
    r = orc.episode_scores(z, z, z[:L], m, np.ones_like(m), np.ones((L, 1), np.int32), l, 3)
    ls = l[:, 0] - np.log(np.exp(l[:, 0].astype(np.float64)).sum(-1, keepdims=True))
    h = (-(np.exp(ls) * ls).sum(-1) / np.log(A))
    np.testing.assert_allclose([r["mean_score"][0], r["max_score"][0]], [h.mean(), h.max()], rtol=1e-5)


def test_segmentation_edge_cases():
    T, B = 6, 4
    z = np.zeros((T + 1, B), np.float32)
    m = np.ones((T + 1, B), np.float32)
    m[0, 0] = 0                       # done at t=0 is skipped (level_sampler.py:504-505)
    m[T, 1] = 0                       # done exactly at T: one full episode, no partial
    m[1:, 2] = 0                      # done every step: T one-step episodes
    recs = orc.episode_scores(z, z, z[:T], m, np.ones_like(m), np.ones((T, B), np.int32), None, 0)
    by = {b: recs[recs["actor"] == b] for b in range(B)}
    assert list(by[0]["num_steps"]) == [T] and list(by[0]["flags"]) == [0]
    assert list(by[1]["num_steps"]) == [T] and list(by[1]["flags"]) == [1]
    assert list(by[2]["num_steps"]) == [1] * T and all(by[2]["flags"] == 1)
    assert list(by[3]["num_steps"]) == [T] and list(by[3]["flags"]) == [0]
    assert list(recs["actor"]) == sorted(recs["actor"])          # actor-major order


def test_regret_modes():
    a = [np.array([0.2, 0.9, 0.0], np.float32), np.array([0.5, 0.9, 0.1], np.float32),
         np.array([0.3, 0.1, 0.0], np.float32), np.array([0.8, 0.4, 0.0], np.float32)]
    np.testing.assert_array_equal(orc.env_return(*a, mode=0), np.maximum(a[3] - a[0], 0))
    np.testing.assert_array_equal(orc.env_return(*a, mode=2), -a[1])
    fp = orc.env_return(*a, mode=1)
    exp = np.where(a[3] > a[1], a[3] - a[0], a[1] - a[2])
    np.testing.assert_array_equal(fp, np.maximum(exp, 0).astype(np.float32))


STAGED = dict(positive_value_loss=1, value_l1=0, signed_value_loss=2, one_step_td_error=6)


@pytest.mark.parametrize("name", ["a0", "a1", "b0", "b1"])
def test_staged_width_rollouts_vs_reference(name):
    """tests/golden/staged.npz: T = 256 / B = 64 and T = 300 / B = 96 (the widths the LDS-staged kernels take), GAE through the
    reference's RolloutStorage and the call sequence of its LevelSampler — pins the oracle at those shapes too."""
    g = np.load(os.path.join(G, "staged.npz"))
    gamma, lam = g["gamma_lambda"]
    r, v, m = g[f"{name}_rewards"], g[f"{name}_value_preds"], g[f"{name}_masks"]
    assert np.array_equal(bits(orc.gae_returns(r, v, m, gamma, lam)[:-1]), bits(g[f"{name}_gae"][:-1]))
    for strategy, code in STAGED.items():
        recs = orc.episode_scores(g[f"{name}_gae"], v, r, m, g[f"{name}_cliffhanger_masks"], g[f"{name}_level_seeds"], None, code, gamma)
        keep = recs[(recs["flags"] & 2) == 0]
        ci, cf = g[f"{name}_{strategy}_calls_i"], g[f"{name}_{strategy}_calls_f"]
        assert len(keep) == len(ci) > 400
        assert np.array_equal(keep["actor"], ci[:, 0]) and np.array_equal(keep["seed"], ci[:, 1])
        assert np.array_equal(keep["num_steps"], ci[:, 2]) and np.array_equal(keep["flags"] & 1, ci[:, 3])
        np.testing.assert_allclose(keep["mean_score"], cf[:, 0], rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(keep["max_score"], cf[:, 1], rtol=1e-6, atol=1e-7)
