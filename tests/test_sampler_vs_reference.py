"""CPU: the host half of LevelSampler / LevelStore against the reference's own outputs
(tests/golden/sampler.npz, store.npz) when fed the oracle's episode records for the same buffers."""
import os

import numpy as np
import pytest

from oracle import orc

G = os.path.join(os.path.dirname(__file__), "golden")
sq = lambda a: a[..., 0]
CODES = {"uniform": 7, "least_confidence": 4, "min_margin": 5, "gae": 2, "value_l1": 0, "signed_value_loss": 2,
         "positive_value_loss": 1, "grounded_signed_value_loss": 8, "grounded_positive_value_loss": 8,
         "one_step_td_error": 6}


class _Space:
    n = 7


def _records(d, returns, strategy):
    return orc.episode_scores(sq(returns), sq(d["value_preds"]), sq(d["rewards"]), sq(d["masks"]),
                              sq(d["cliffhanger_masks"]), sq(d["level_seeds"]), d["action_log_dist"],
                              CODES[strategy], 0.995)


@pytest.mark.parametrize("strategy", sorted(CODES))
def test_fixed_seed_update_matches_reference(strategy):
    from level_replay import LevelSampler
    g, s = np.load(os.path.join(G, "scans.npz")), np.load(os.path.join(G, "sampler.npz"))
    d = {k[len("small_"):]: g[k] for k in g.files if k.startswith("small_")}
    B = d["rewards"].shape[1]
    ls = LevelSampler(list(range(1, 12)), None, _Space(), num_actors=B, strategy=strategy, score_transform="rank",
                      temperature=0.1, staleness_coef=0.3, rho=0.5, replay_prob=0.5, alpha=1.0, gamma=0.995)
    ls.update_with_records(_records(d, d["gae"], strategy), values=sq(d["value_preds"]))
    tol = dict(rtol=3e-6, atol=3e-7)
    np.testing.assert_allclose(ls.seed_scores, s[f"{strategy}_seed_scores"], **tol)
    np.testing.assert_allclose(ls.partial_seed_scores, s[f"{strategy}_partial_scores"], **tol)
    np.testing.assert_allclose(ls.partial_seed_max_scores, s[f"{strategy}_partial_max"], **tol)
    assert np.array_equal(ls.partial_seed_steps, s[f"{strategy}_partial_steps"])
    assert np.array_equal(ls.unseen_seed_weights, s[f"{strategy}_unseen"])
    if ls.grounded_values is not None:
        np.testing.assert_allclose(ls.grounded_values, s[f"{strategy}_grounded"], **tol)
    if strategy != "uniform":                    # rank transform of tied scores depends on argsort tie order
        np.testing.assert_allclose(ls.sample_weights(), s[f"{strategy}_weights"], rtol=1e-9)
    ls.after_update()
    np.testing.assert_allclose(ls.seed_scores, s[f"{strategy}_seed_scores_after"], **tol)
    assert not ls.partial_seed_scores.any() and not ls.partial_seed_steps.any()


@pytest.mark.parametrize("transform", ["constant", "eps_greedy", "rank", "power", "softmax", "match", "match_rank"])
@pytest.mark.parametrize("staleness", [0.0, 0.3])
def test_score_transforms_match_reference(transform, staleness):
    from level_replay import LevelSampler
    s = np.load(os.path.join(G, "sampler.npz"))
    ls = LevelSampler(list(range(1, 12)), None, _Space(), num_actors=8, strategy="value_l1", score_transform=transform,
                      temperature=0.3, staleness_coef=staleness, staleness_transform="power", staleness_temperature=1.0)
    ls.seed_scores, ls.seed_staleness, ls.unseen_seed_weights = s["tf_scores"].copy(), s["tf_stale"].copy(), s["tf_unseen"].copy()
    np.testing.assert_allclose(ls.sample_weights(), s[f"tf_{transform}_{staleness}"], rtol=1e-12, atol=0)


def test_buffer_mode_sequence_matches_reference():
    """Robust-PLR style staging -> working buffer with eviction, staleness and solvability, 4 updates."""
    from level_replay import LevelSampler
    s = np.load(os.path.join(G, "sampler.npz"))
    B = s["buf0_rewards"].shape[1]
    ls = LevelSampler([], None, _Space(), num_actors=B, strategy="positive_value_loss", score_transform="rank",
                      temperature=0.1, staleness_coef=0.3, rho=0.5, replay_prob=0.5, alpha=1.0,
                      sample_full_distribution=True, seed_buffer_size=6, seed_buffer_priority="replay_support", gamma=0.995)
    for it in range(4):
        d = {k[len(f"buf{it}_"):]: s[k] for k in s.files if k.startswith(f"buf{it}_")}
        ls.observe_external_unseen_sample([int(x) for x in d["observed"]], [bool(x) for x in d["solvable"]])
        ls.update_with_records(_records(d, d["returns"], "positive_value_loss"))
        ls.after_update()
        assert np.array_equal(ls.seeds, d["seeds"]), it
        np.testing.assert_allclose(ls.seed_scores, d["seed_scores"], rtol=3e-6, atol=3e-7)
        assert np.array_equal(ls.seed_staleness, d["staleness"]) and np.array_equal(ls.unseen_seed_weights, d["unseen"])
        np.testing.assert_allclose(ls.sample_weights(), d["weights"], rtol=1e-9)
        np.testing.assert_allclose(ls.solvable_mass, d["solvable_mass"], rtol=1e-9)
        assert ls._proportion_filled == d["filled"] and len(ls.staging_seed_set) == s["buf_staging_sizes"][it]


def test_replay_decisions_and_sampling_follow_numpy_stream():
    from level_replay import LevelSampler
    ls = LevelSampler(list(range(20)), None, _Space(), num_actors=4, strategy="value_l1", score_transform="rank",
                      temperature=0.3, rho=0.2, replay_prob=0.7, staleness_coef=0.1)
    ls.seed_scores = np.linspace(0, 1, 20)
    ls.unseen_seed_weights[:10] = 0
    np.random.seed(3)
    dec = [ls.sample_replay_decision() for _ in range(50)]
    np.random.seed(3)
    assert dec == [bool(np.random.rand() < 0.7) for _ in range(50)]
    np.random.seed(4)
    picks = ls.sample_replay_levels(200)
    assert set(picks) <= set(range(10)) and ls.seed_staleness[picks[-1]] == 0


def test_level_store_matches_reference():
    from level_replay import LevelStore
    s = np.load(os.path.join(G, "store.npz"))
    st = LevelStore()
    a = st.insert(["1 2 3", "4 5 6", "1 2 3"])
    b = st.insert(["7 8 9", "4 5 6"], parent_seeds=[a[0], a[1]])
    assert np.array_equal(a, s["ins_a"]) and np.array_equal(b, s["ins_b"])
    assert np.array_equal([len(st.seed2parent[k]) for k in sorted(st.seed2parent)], s["parents_len"])
    st.reconcile_seeds({1, 3})
    assert np.array_equal(sorted(st.seed2level), s["after_reconcile"])
    enc = np.arange(27, dtype=np.uint8).reshape(3, 3, 3)
    st2 = LevelStore(data_info={"numpy": True, "dtype": enc.dtype, "shape": enc.shape})
    sid = st2.insert([enc.tobytes()])
    assert np.array_equal(st2.get_level(sid[0]), s["bytes_roundtrip"])
