"""-m gpu: the HIP MultiGrid engine (through the C ABI) against the CPU oracle, bit-exact.

Every output tensor of every call (u8 and fp32 observations, reward, done, info bits, episode
stats, masks, truncated obs, teacher obs, random_z) and the full state after each call
(encodings, agent/level scalars, metrics, rng counters, error words) must be identical on
identical seeds — integer work and fp32 values produced by identical operations.
"""
import numpy as np
import pytest

from oracle import orc
from tests import mg_diff

pytestmark = pytest.mark.gpu

CONFIGS = [
    # BASELINE config: MultiGrid-GoalLastFewerBlocksAdversarial-v0 (adversarial.py:614-618)
    (0, dict(n_clutter=25, choose_goal_last=1), False, 130),
    (1, dict(size=6, n_clutter=7, max_steps=50, max_episode_steps=50, choose_goal_last=1), False, 70),
    (2, dict(n_clutter=25, choose_goal_last=1, max_episode_steps=50), True, 64),
    (3, dict(resample_n_clutter=1, n_clutter=60, choose_goal_last=1, editor_actions=1), False, 65),
    (4, dict(choose_goal_last=0, goal_noise=0.3, n_clutter=50, editor_actions=1), True, 63),
    (5, dict(size=10, n_clutter=30, max_steps=200, max_episode_steps=200, agent_view_size=7, editor_actions=0), True, 1),
    (6, dict(n_clutter=25, choose_goal_last=1, agent_view_size=3, fixed_environment=1), False, 200),
    # opaque walls (Grid.process_vis; MultiGrid-GoalLast[FewerBlocks]OpaqueWallsAdversarial-v0, adversarial.py:609-612,655-658)
    (7, dict(n_clutter=25, choose_goal_last=1, see_through_walls=0), False, 130),
    (8, dict(n_clutter=50, choose_goal_last=1, see_through_walls=0, max_episode_steps=40), True, 67),
    (9, dict(size=10, n_clutter=30, max_steps=200, max_episode_steps=60, agent_view_size=7, see_through_walls=0), True, 33),
    (10, dict(size=6, n_clutter=7, max_steps=50, max_episode_steps=50, agent_view_size=3, see_through_walls=0), False, 100),
]


@pytest.mark.parametrize("seed,kw,exotic,B", CONFIGS)
def test_engine_matches_oracle(seed, kw, exotic, B):
    from tests.hip_adapter import HipMg
    cfg = dict(num_envs=B, engine_seed=seed, env_index_base=1000 * seed)
    cfg.update(kw)
    x = orc.OracleMg(**cfg)
    y = HipMg(**cfg)
    rng = np.random.default_rng(seed)
    mg_diff.compare_state(x, y, "after create")
    episodes = mg_diff.drive(x, y, rng, n_calls=24, steps_per_call=25, exotic=exotic, check_every=5)
    assert episodes >= 0


def test_paired_shape_rollout_bitexact():
    """4096 envs, teacher builds levels with random actions, 256 student steps with Philox actions:
    per-step outputs and final state identical to the oracle (BASELINE configs[1] shape)."""
    from tests.hip_adapter import HipMg
    import torch
    import dcd_hip
    B, T = 4096, 256
    cfg = dict(num_envs=B, n_clutter=25, choose_goal_last=1, engine_seed=7)
    x = orc.OracleMg(**cfg)
    y = HipMg(**cfg)
    rng = np.random.default_rng(7)
    x.reset(); y.reset()
    for k in range(27):
        loc = rng.integers(0, 169, B)
        ox, oy = x.step_adversary(loc), y.step_adversary(loc)
    mg_diff.assert_same(ox, oy, "teacher last obs")
    mg_diff.assert_same(x.reset_agent(), y.reset_agent(), "reset_agent")
    tx = ty = None
    for t in range(T):
        a = orc.random_actions(B, 0, 1, t, 7)
        a_dev = dcd_hip.engine.random_actions(B, 0, 1, t, 7, "cuda:0").cpu().numpy()
        assert np.array_equal(a, a_dev)
        flags = 2 if t == T - 1 else 0
        ox, oy = x.step(a, flags, tx), y.step(a, flags, ty)
        tx, ty = ox["term_obs"], oy["term_obs"]
        if t % 16 == 0 or t == T - 1:
            mg_diff.assert_same(ox, oy, f"step {t}")
    mg_diff.compare_state(x, y, "after rollout")


@pytest.mark.parametrize("B,kw", [(130, dict(n_clutter=25, choose_goal_last=1, max_episode_steps=30)),
                                  (4096, dict(n_clutter=25, choose_goal_last=1)),
                                  (77, dict(size=10, n_clutter=30, max_steps=200, max_episode_steps=40, agent_view_size=7, see_through_walls=0)),
                                  (8192 + 37, dict(n_clutter=25, choose_goal_last=1, max_episode_steps=25)),
                                  (16384 + 5, dict(n_clutter=25, choose_goal_last=1, max_episode_steps=25))])
def test_step_pair_matches_oracle(B, kw):
    """dcd_mg_step_pair: two replicas of the batch (the protagonist's and the antagonist's copy of the teacher's levels)
    advanced in ONE launch with different actions == two oracle batches stepped one after the other.  Every output
    and the full state of both, with ragged last workgroups, all three workgroup widths, the reset-random and
    last-step variants, and level ops applied to one replica only in between."""
    from tests.hip_adapter import HipMg
    cfg = dict(num_envs=B, engine_seed=21, env_index_base=100)
    cfg.update(kw)
    x0, x1 = orc.OracleMg(**cfg), orc.OracleMg(**cfg)
    y0 = HipMg(replicas=2, **cfg)
    y1 = y0.replica(1)
    rng = np.random.default_rng(B)
    mg_diff.compare_state(x1, y1, "replica after create")
    n_adv = cfg["n_clutter"] + 2
    interior = (x0.S - 2) ** 2
    x0.reset(); y0.reset()
    for k in range(n_adv):
        loc = rng.integers(0, interior, B)
        x0.step_adversary(loc); y0.step_adversary(loc)
    x1.clone_levels(x0); y1.clone_levels(y0)
    mg_diff.compare_state(x1, y1, "clone")
    mg_diff.assert_same(x0.reset_agent(), y0.reset_agent(), "reset_agent 0")
    mg_diff.assert_same(x1.reset_agent(), y1.reset_agent(), "reset_agent 1")
    t0 = t1 = u0 = u1 = None
    T = 60 if B <= 4096 else 30
    for t in range(T):
        a0 = rng.choice(7, size=B, p=[0.15, 0.15, 0.5, 0.05, 0.05, 0.05, 0.05])
        a1 = rng.choice(7, size=B, p=[0.2, 0.2, 0.4, 0.05, 0.05, 0.05, 0.05])
        flags = (1 if t % 7 == 3 else 0) | (2 if t == T - 1 else 0)
        o0, o1 = x0.step(a0, flags, t0), x1.step(a1, flags, t1)
        p0, p1 = y0.step_pair(y1, a0, a1, flags, u0, u1)
        t0, t1, u0, u1 = o0["term_obs"], o1["term_obs"], p0["term_obs"], p1["term_obs"]
        mg_diff.assert_same(o0, p0, f"pair step {t} replica 0")
        mg_diff.assert_same(o1, p1, f"pair step {t} replica 1")
        if t == T // 2:                                   # a level op on ONE replica: the other must not move
            enc = mg_diff.random_encodings(rng, B, x0.S, exotic=True)
            mg_diff.assert_same(x1.reset_to_encoding(enc), y1.reset_to_encoding(enc), "encoding on replica 1")
        if t % 10 == 0 or t == T - 1:
            mg_diff.compare_state(x0, y0, f"pair step {t} replica 0")
            mg_diff.compare_state(x1, y1, f"pair step {t} replica 1")


@pytest.mark.parametrize("B,kw", [(4096, dict(n_clutter=25, choose_goal_last=1, max_episode_steps=40)),
                                  (130, dict(n_clutter=25, choose_goal_last=1, max_episode_steps=30, see_through_walls=0)),
                                  (77, dict(size=10, n_clutter=30, max_steps=200, max_episode_steps=40, agent_view_size=7)),
                                  (200, dict(size=6, n_clutter=7, max_steps=50, max_episode_steps=50, agent_view_size=3)),
                                  (32768 + 64, dict(n_clutter=25, choose_goal_last=1, max_episode_steps=25))])
def test_u8_only_emission_matches_oracle(B, kw):
    """The step kernel's raw-u8-only variant (what RolloutStorage's u8 mode asks for): observations staged in LDS and
    written as 16-byte blocks, no fp32 tables — single and paired launches, ragged last workgroups, every view size,
    opaque walls, truncated observations, every workgroup width."""
    import torch
    import dcd_hip
    from tests.hip_adapter import _np
    cfg = dict(num_envs=B, engine_seed=4, env_index_base=50)
    cfg.update(kw)
    x0, x1 = orc.OracleMg(**cfg), orc.OracleMg(**cfg)
    eng_cfg = {k: v for k, v in cfg.items() if k != "num_envs"}
    y0 = dcd_hip.MultiGridEngine(B, replicas=2, **eng_cfg)
    y1 = y0.replica(1)
    rng = np.random.default_rng(B)
    for e in (x0, x1):
        e.set_seed(np.arange(B) + 7)
    for e in (y0, y1):
        e.set_seed(torch.arange(B) + 7)
    enc = mg_diff.random_encodings(rng, B, x0.S, exotic=True)
    for e in (x0, x1):
        e.reset_to_encoding(enc)
    for e in (y0, y1):
        e.reset_to_encoding(torch.from_numpy(enc).to("cuda:0"))
    t0 = t1 = None
    u0, u1 = y0.step_buffers(u8=True, f32=False), y1.step_buffers(u8=True, f32=False)
    dev = lambda a: torch.as_tensor(a, dtype=torch.int64).to("cuda:0")
    for t in range(50):
        a0 = rng.choice(7, size=B, p=[0.15, 0.15, 0.5, 0.05, 0.05, 0.05, 0.05])
        a1 = rng.choice(7, size=B, p=[0.2, 0.2, 0.4, 0.05, 0.05, 0.05, 0.05])
        o0, o1 = x0.step(a0, 0, t0), x1.step(a1, 0, t1)
        t0, t1 = o0["term_obs"], o1["term_obs"]
        if t % 2:
            y0.step_pair(y1, dev(a0), dev(a1), 0, u0, u1)
        else:
            y0.step(dev(a0), 0, u0)
            y1.step(dev(a1), 0, u1)
        for o, u in ((o0, u0), (o1, u1)):
            g = _np(u)
            for k in ("image_u8", "dir_u8"):
                assert np.array_equal(g["obs"][k], o["obs"][k]), (t, k)
                assert np.array_equal(g["term_obs"][k], o["term_obs"][k]), (t, "term", k)
            for k in ("reward", "done", "info", "ep_ret", "ep_len", "masks"):
                assert np.array_equal(g[k], o[k]), (t, k)
    assert np.array_equal(y0.get_encodings().cpu().numpy(), x0.get_encodings())
    assert np.array_equal(y1.get_state().cpu().numpy(), x1.get_state())


def test_fused_random_steps_match_single_steps():
    """dcd_mg_step_random(K) == K single dcd_mg_step calls with the same Philox actions."""
    import torch
    import dcd_hip
    B, K = 300, 40
    cfg = dict(size=15, n_clutter=25, choose_goal_last=1, engine_seed=3)
    a = dcd_hip.MultiGridEngine(B, **cfg)
    b = dcd_hip.MultiGridEngine(B, **cfg)
    for e in (a, b):
        e.reset_random()
    n_eps = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    out_a = a.step_buffers()
    a.step_random(11, 5, K, 0, out_a, n_eps)
    eps = 0
    out_b = None
    for k in range(K):
        act = dcd_hip.engine.random_actions(B, 0, 11, 5 + k, 7, "cuda:0")
        out_b = b.step(act, 0, out_b)
        eps += int((out_b["info"] & 1).sum())
    assert torch.equal(a.get_encodings(), b.get_encodings())
    assert torch.equal(a.get_state(), b.get_state())
    for k in ("reward", "done", "info", "ep_ret", "ep_len"):
        assert torch.equal(out_a[k], out_b[k]), k
    assert torch.equal(out_a["obs"]["image_f32"], out_b["obs"]["image_f32"])
    assert int(n_eps.item()) == eps


def test_error_latch():
    import torch
    import dcd_hip
    e = dcd_hip.MultiGridEngine(8, n_clutter=3, choose_goal_last=1)
    e.reset()
    bad = torch.full((8,), 169, dtype=torch.int64, device="cuda:0")
    e.step_adversary(bad)
    with pytest.raises(ValueError, match="outside the grid"):
        e.raise_errors()
    assert e.poll_errors() == (0, -1)
    e.reset()
    e.reset_agent()                     # no start position yet
    with pytest.raises(ValueError, match="empty start position"):
        e.raise_errors()
    with pytest.raises(dcd_hip.DcdHipError):
        e.step(torch.zeros(8, dtype=torch.int64))      # CPU tensor must be refused, not copied


@pytest.mark.parametrize("B,seed", [(16384 + 37, 11), (32768 + 5, 12)])
def test_wide_workgroup_variants_match_oracle(B, seed):
    """The step / level kernels pick 32 or 64 envs per workgroup from B >= 16 384 / 32 768 (the HBM-bound regime of the
    roofline sweep); the randomised driver at those batch sizes, with a ragged last workgroup."""
    from tests.hip_adapter import HipMg
    cfg = dict(num_envs=B, engine_seed=seed, env_index_base=5 * seed, n_clutter=25, choose_goal_last=1, max_episode_steps=40)
    x, y = orc.OracleMg(**cfg), HipMg(**cfg)
    mg_diff.compare_state(x, y, "after create")
    mg_diff.drive(x, y, np.random.default_rng(seed), n_calls=9, steps_per_call=12, exotic=True, check_every=3)
