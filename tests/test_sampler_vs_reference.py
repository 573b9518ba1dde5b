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


@pytest.mark.parametrize("score_transform,staleness_coef,staleness_transform", [
    ("rank", 0.3, "power"), ("rank", 0.0, "power"), ("power", 0.3, "power"), ("softmax", 0.5, "rank"),
    ("eps_greedy", 0.3, "power"), ("match_rank", 0.1, "power"), ("max", 0.3, "power")])
def test_batched_replay_sampling_equals_sequential_calls(score_transform, staleness_coef, staleness_transform):
    """sample_replay_levels(n) (one pass over the unchanging score weights, hand-rolled inverse-CDF draw) is the same
    function as n calls of sample_replay_level(): same seeds, same global-RNG state afterwards, same staleness."""
    from level_replay import LevelSampler

    class Space:
        n = 7

    def make():
        s = LevelSampler([], None, Space(), num_actors=4, strategy="positive_value_loss", score_transform=score_transform,
                         temperature=0.3, staleness_coef=staleness_coef, staleness_transform=staleness_transform,
                         staleness_temperature=1.0, sample_full_distribution=True, seed_buffer_size=300, rho=0.5, replay_prob=0.5)
        rng = np.random.default_rng(5)
        N = 300
        s.seeds[:] = np.arange(11, N + 11)
        s.seed2index = {int(sd): i for i, sd in enumerate(s.seeds)}
        s.seed_scores[:] = rng.random(N) * (rng.random(N) > 0.1)
        s.unseen_seed_weights[:] = (rng.random(N) > 0.8).astype(np.float64)
        s.seed_staleness[:] = rng.integers(0, 40, N)
        s.working_seed_buffer_size = N
        return s

    a, b = make(), make()
    np.random.seed(123)
    seq = [a.sample_replay_level() for _ in range(200)]
    state_a = np.random.get_state()[1].copy()
    np.random.seed(123)
    bat = b.sample_replay_levels(200)
    state_b = np.random.get_state()[1].copy()
    assert seq == bat
    assert np.array_equal(state_a, state_b)
    assert np.array_equal(a.seed_staleness, b.seed_staleness)
    assert all(a.unseen_seed_weights[a.seed2index[sd]] == 0 for sd in bat)      # only seen levels are replayed


def test_native_draw_loop_is_numpy_bit_for_bit(monkeypatch):
    """dcd_sampler_draws / dcd_np_sum (csrc/host_utils.cpp, host memory): numpy's pairwise np.sum reproduced exactly,
    and the C draw loop == the numpy draw loop == sequential sample_replay_level() calls (seeds, staleness, RNG state),
    including a buffer with unseen slots; with the native loop switched off the numpy loop gives the same again."""
    import ctypes as C
    import level_replay.level_sampler as ls
    from dcd_hip import lib
    from level_replay import LevelSampler
    L = lib()
    L.dcd_np_sum.restype = C.c_double
    rng = np.random.default_rng(0)
    for n in list(range(1, 140)) + [255, 256, 257, 1000, 4000, 4001, 9999]:
        a = rng.random(n) * rng.choice([1.0, 1e-6, 1e6])
        assert L.dcd_np_sum(a.ctypes.data_as(C.c_void_p), C.c_int64(n)) == float(np.sum(a)), n
    assert ls._native_draws_available()

    class Space:
        n = 7

    def make():
        s = LevelSampler([], None, Space(), num_actors=4, strategy="positive_value_loss", score_transform="rank", temperature=0.3,
                         staleness_coef=0.3, sample_full_distribution=True, seed_buffer_size=500, rho=0.5, replay_prob=0.5)
        r = np.random.default_rng(9)
        s.seeds[:] = np.arange(5, 505)
        s.seed2index = {int(sd): i for i, sd in enumerate(s.seeds)}
        s.seed_scores[:] = r.random(500) * (r.random(500) > 0.2)
        s.unseen_seed_weights[:] = (r.random(500) > 0.85).astype(np.float64)
        s.seed_staleness[:] = r.integers(0, 25, 500)
        s.working_seed_buffer_size = 500
        return s

    results = []
    for mode in ("sequential", "native", "numpy"):
        s = make()
        monkeypatch.setattr(ls, "_NATIVE_DRAWS", mode == "native")
        np.random.seed(77)
        seeds = [s.sample_replay_level() for _ in range(300)] if mode == "sequential" else s.sample_replay_levels(300)
        results.append((seeds, s.seed_staleness.copy(), np.random.get_state()[1].copy()))
    for seeds, stale, state in results[1:]:
        assert seeds == results[0][0] and np.array_equal(stale, results[0][1]) and np.array_equal(state, results[0][2])


def test_native_draw_loop_paths_and_early_stops():
    """dcd_sampler_draws has two paths — integer staleness counts with a 0/1 mask (running integer sums, lazy ageing) and the
    general one (three passes) — both must equal the numpy statement draw for draw and leave the staleness array as numpy
    would, also when a draw cannot be made (all staleness mass on unseen slots: the loop stops and reports how far it got) and
    when the call is split into several calls (a replay rollout draws a few levels per step)."""
    import ctypes as C
    from dcd_hip import lib
    L = lib()
    L.dcd_sampler_draws.restype = C.c_int64
    p_ = lambda a: a.ctypes.data_as(C.c_void_p)

    def numpy_draws(wsc, c, st, keep, u):
        st, out = st.copy(), []
        for uj in u:
            sw = st * keep
            z = np.sum(sw)
            if not z > 0:
                break
            w = wsc + c * (sw / z)
            tot = np.sum(w)
            w = w / tot if tot != 1.0 else w
            cdf = np.cumsum(w)
            cdf /= cdf[-1]
            i = int(np.searchsorted(cdf, uj, side="right"))
            out.append(i)
            st = st + 1
            st[i] = 0
        return out, st
    rng = np.random.default_rng(3)
    for N, n, integral in ((777, 300, True), (777, 300, False), (130, 64, True), (4000, 50, True)):
        wsc = rng.random(N)
        wsc = 0.7 * wsc / wsc.sum()
        keep = (rng.random(N) < 0.9).astype(np.float64)
        st0 = rng.integers(0, 40, N).astype(np.float64) + (0.0 if integral else 0.25)
        u = rng.random(n)
        want, st_want = numpy_draws(wsc, 0.3, st0, keep, u)
        st, idx = st0.copy(), np.zeros(n, dtype=np.int64)
        k = L.dcd_sampler_draws(C.c_int64(N), C.c_int64(n), p_(wsc), C.c_double(0.3), p_(st), p_(keep), p_(u), C.c_int32(1), p_(idx))
        assert k == n and idx.tolist() == want and np.array_equal(st, st_want), (N, integral)
        st, got = st0.copy(), []                                     # the same draws in calls of 7
        for j in range(0, n, 7):
            m = min(7, n - j)
            part, uu = np.zeros(m, dtype=np.int64), np.ascontiguousarray(u[j:j + m])
            assert L.dcd_sampler_draws(C.c_int64(N), C.c_int64(m), p_(wsc), C.c_double(0.3), p_(st), p_(keep), p_(uu), C.c_int32(1), p_(part)) == m
            got += part.tolist()
        assert got == want and np.array_equal(st, st_want), (N, integral, "split")
        st, idx = st0.copy(), np.zeros(n, dtype=np.int64)            # update_staleness = 0: the same distribution every draw
        L.dcd_sampler_draws(C.c_int64(N), C.c_int64(5), p_(wsc), C.c_double(0.3), p_(st), p_(keep), p_(u), C.c_int32(0), p_(idx))
        assert np.array_equal(st, st0) and idx[0] == want[0]
    # staleness mass only where keep == 0: the first draw works (mass on seen slots), afterwards every seen slot... stops at z == 0
    N = 50
    keep = np.zeros(N); keep[:3] = 1.0
    st0 = np.zeros(N); st0[:3] = 0.0; st0[3:] = 9.0
    wsc = np.full(N, 0.7 / N)
    u = rng.random(4)
    st, idx = st0.copy(), np.zeros(4, dtype=np.int64)
    k = L.dcd_sampler_draws(C.c_int64(N), C.c_int64(4), p_(wsc), C.c_double(0.3), p_(st), p_(keep), p_(u), C.c_int32(1), p_(idx))
    assert k == 0 and np.array_equal(st, st0)                          # z = 0 at the first draw: nothing drawn, nothing aged


@pytest.mark.parametrize("name", ["a0", "a1", "b0", "b1"])
@pytest.mark.parametrize("strategy", ["positive_value_loss", "one_step_td_error", "grounded_positive_value_loss", "grounded_signed_value_loss"])
def test_staged_width_rollouts_sampler_state_matches_reference(name, strategy):
    """tests/golden/staged.npz (T = 256 / B = 64, T = 300 / B = 96; "…1": envs whose rollout ends mid-episode): sampler state after
    update_with_rollouts, incl. the grounded strategies' host-side advantage and the trailing-partial-segment rule."""
    from level_replay import LevelSampler
    g = np.load(os.path.join(G, "staged.npz"))
    d = {k: g[f"{name}_{k}"] for k in ("rewards", "value_preds", "masks", "cliffhanger_masks", "level_seeds", "gae")}
    B = d["rewards"].shape[1]
    recs = orc.episode_scores(d["gae"], d["value_preds"], d["rewards"], d["masks"], d["cliffhanger_masks"], d["level_seeds"], None,
                              CODES[strategy], float(g["gamma_lambda"][0]))
    ls = LevelSampler(list(range(1, 12)), None, _Space(), num_actors=B, strategy=strategy, score_transform="rank", temperature=0.1,
                      staleness_coef=0.3, rho=0.5, replay_prob=0.5, alpha=1.0, gamma=float(g["gamma_lambda"][0]))
    ls.update_with_records(recs, values=d["value_preds"])
    np.testing.assert_allclose(ls.seed_scores, g[f"{name}_{strategy}_seed_scores"], rtol=3e-6, atol=3e-7)
    if ls.grounded_values is not None:
        np.testing.assert_allclose(ls.grounded_values, g[f"{name}_{strategy}_grounded"], rtol=3e-6, atol=3e-7)
