"""CPU: properties that pin oracle/mg_oracle.c to the reference where the reference offers a pin
(SURVEY §8c): the closed-form view map (multigrid.py:1080-1093) vs the slice+rotate render
(:990-996), the reference's inline asserts, constants implied by the repo, Philox known answers."""
import numpy as np
import pytest

from oracle import orc
from tests import mg_diff


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_obs_scale_is_fp64_division():
    # obs_wrappers.py:90-91: image/10.0 in numpy fp64, then .float()
    m = orc.OracleMg(num_envs=1)
    enc = mg_diff.random_encodings(np.random.default_rng(0), 1, 15, exotic=True)
    o = m.reset_to_encoding(enc)
    exp = (o["image_u8"].astype(np.float64) / 10.0).transpose(0, 3, 1, 2).astype(np.float32)
    assert np.array_equal(o["image_f32"], exp)
    assert np.array_equal(o["dir_f32"][:, 0], o["dir_u8"].astype(np.float32))


@pytest.mark.parametrize("V", [3, 5, 7])
def test_view_render_matches_closed_form_map(V):
    """gen_obs (slice + rotate_left × (dir+1)) against the world<->view map of
    compute_agent_visibility_mask — two independent statements of the same geometry."""
    rng = np.random.default_rng(V)
    B, S = 64, 15
    m = orc.OracleMg(num_envs=B, agent_view_size=V)
    enc = mg_diff.random_encodings(rng, B, S, exotic=True)
    m.reset_to_encoding(enc)
    for t in range(40):
        out = m.step(rng.integers(0, 3, B))
        grid = m.get_encodings()
        st = m.get_state()
        img = out["obs"]["image_u8"]
        for b in range(0, B, 7):
            ax, ay, d = st[b, 0], st[b, 1], st[b, 2]
            carrying = st[b, 3]
            for vx in range(V):
                for vy in range(V):
                    wx, wy = orc.view_to_world(int(ax), int(ay), int(d), V, vx, vy)
                    got = tuple(img[b, vx, vy])
                    if vx == V // 2 and vy == V - 1:
                        exp = (1, 0, 0) if carrying == 0 else (carrying & 255, (carrying >> 8) & 255, carrying >> 16)
                    elif not (0 <= wx < S and 0 <= wy < S):
                        exp = (2, 5, 0)
                    else:
                        exp = tuple(grid[b, wx, wy])
                    assert got == exp, (b, t, vx, vy, got, exp)


def test_teacher_action_to_cell_map_and_order():
    # adversarial.py:476-482: x = loc % 13 + 1, y = loc // 13 + 1; goal at step n-2, agent at n-1
    m = orc.OracleMg(num_envs=1, n_clutter=3, choose_goal_last=1)
    m.reset()
    locs = [0, 14, 168, 30, 31]          # walls at (1,1),(2,2),(13,13); goal (5,3); agent (6,3)
    for k, l in enumerate(locs):
        o = m.step_adversary([l])
        assert o["time_step"][0, 0] == k + 1
        assert o["done"][0] == (k == 4)
    g = m.get_encodings()[0]
    assert tuple(g[1, 1]) == (2, 5, 0) and tuple(g[2, 2]) == (2, 5, 0) and tuple(g[13, 13]) == (2, 5, 0)
    assert tuple(g[5, 3]) == (8, 1, 0) and g[6, 3, 0] == 10 and g[6, 3, 1] == 0
    met = m.get_metrics()
    assert met["n_clutter_placed"][0] == 3 and met["passable"][0] == 1
    assert met["shortest_path_length"][0] == 1 and met["distance_to_goal"][0] == 1


def test_unreachable_goal_has_path_length_170():
    m = orc.OracleMg(num_envs=1, n_clutter=8, choose_goal_last=1)
    m.reset()
    ring = [(5, 5), (6, 5), (7, 5), (5, 6), (7, 6), (5, 7), (6, 7), (7, 7)]      # walls around (6,6)
    for (x, y) in ring:
        m.step_adversary([(y - 1) * 13 + (x - 1)])
    m.step_adversary([(6 - 1) * 13 + (6 - 1)])      # goal inside the ring
    m.step_adversary([0])                            # agent at (1,1)
    met = m.get_metrics()
    assert met["passable"][0] == 0 and met["shortest_path_length"][0] == 170


def test_wall_on_occupied_cell_is_noop_and_agent_on_goal_goes_random():
    m = orc.OracleMg(num_envs=1, n_clutter=2, choose_goal_last=1)
    m.reset()
    m.step_adversary([5]); m.step_adversary([5])     # second wall on the same cell: nothing
    assert m.get_metrics()["n_clutter_placed"][0] == 1
    m.step_adversary([5])                            # goal replaces the wall (remove_wall + put_obj)
    m.step_adversary([5])                            # agent on goal -> random free cell
    st, g = m.get_state()[0], m.get_encodings()[0]
    assert tuple(g[6, 1]) == (8, 1, 0) and (st[8], st[9]) != (6, 1) and g[st[8], st[9], 0] == 10
    assert m.get_metrics()["n_clutter_placed"][0] == 0


def test_step_invariants_and_quirks():
    """Reference asserts as properties (multigrid.py:485-491,844-853): one agent object in the grid,
    at agent_pos, facing agent_dir; moving erases the cell it leaves (A.3-1); goal stays (A.3-2)."""
    rng = np.random.default_rng(1)
    B = 32
    m = orc.OracleMg(num_envs=B, size=8, n_clutter=6, max_steps=40, max_episode_steps=40, choose_goal_last=1)
    m.reset_random()
    goals = m.get_state()[:, 11:13].copy()
    eps = 0
    for t in range(400):
        out = m.step(rng.choice(7, size=B, p=[0.2, 0.2, 0.5, 0.025, 0.025, 0.025, 0.025]))
        g, st = m.get_encodings(), m.get_state()
        assert ((g[..., 0] == 10).sum(axis=(1, 2)) == 1).all()
        for b in range(B):
            assert g[b, st[b, 0], st[b, 1], 0] == 10 and g[b, st[b, 0], st[b, 1], 2] == st[b, 2]
            assert tuple(g[b, goals[b, 0], goals[b, 1]]) == (8, 1, 0)
        done = out["done"].astype(bool)
        assert (st[done, 4] == 0).all() and (st[done, 5] == 0).all()          # auto reset
        assert ((out["reward"] > 0) <= done).all()
        # reward LUT: 1 - 0.9 * step/max_steps in fp64, rounded once
        r = out["reward"][out["reward"] > 0]
        l = out["ep_len"][out["reward"] > 0]
        assert np.array_equal(r, (1.0 - 0.9 * (l / 40.0)).astype(np.float32))
        eps += done.sum()
    assert eps > 100 and m.poll_errors() == (0, -1)


def test_time_limit_flags():
    m = orc.OracleMg(num_envs=1, size=15, n_clutter=0, max_steps=250, max_episode_steps=10, choose_goal_last=1)
    m.reset(); m.step_adversary([100]); m.step_adversary([0]); m.reset_agent()
    for t in range(10):
        out = m.step([0])                         # spin in place
    # TimeLimit fires first: truncated=True, truncated_obs present, done (time_limit.py:28-31)
    assert out["done"][0] == 1 and out["info"][0] == (1 | 2 | 4) and out["bad_masks"][0] == 0
    assert out["ep_len"][0] == 10 and out["term_obs"]["image_u8"].any()
    out = m.step([0], flags=2)                    # last rollout step: cliffhanger (runner :521-530)
    assert out["done"][0] == 1 and out["info"][0] == (2 | 4 | 8) and out["masks"][0] == 0
    assert out["cliffhanger_masks"][0] == 0 and out["bad_masks"][0] == 0


def test_errors_are_latched():
    m = orc.OracleMg(num_envs=3, n_clutter=1, choose_goal_last=1)
    m.reset()
    m.step_adversary([0, 169, 5])
    assert m.poll_errors() == (2, 1) and m.poll_errors() == (0, -1)
    m.reset(); m.reset_agent()
    assert m.poll_errors()[0] == 4
    enc = mg_diff.random_encodings(np.random.default_rng(0), 3, 15)
    enc[1, 3, 3] = (11, 0, 0)
    m.reset_to_encoding(enc)
    assert m.poll_errors() == (8, 1)


# ---- opaque walls: Grid.process_vis known answers (hand-derived from the algorithm text) ----------
def _opaque_view(cls, walls, V=5):
    """15x15 walled grid, agent at (7,7), goal at (1,1), `walls` extra wall cells; returns the type channel of the
    view as rows vy = 0 (far) .. V-1 (agent's row), columns vx, for whatever start direction was drawn, plus dir."""
    o = cls(num_envs=1, size=15, agent_view_size=V, see_through_walls=0, n_clutter=25, choose_goal_last=1)
    enc = np.zeros((1, 15, 15, 3), np.uint8)
    enc[..., 0] = 1
    for k in range(15):
        for (x, y) in ((0, k), (14, k), (k, 0), (k, 14)):
            enc[0, x, y] = (2, 5, 0)
    for (x, y) in walls:
        enc[0, x, y] = (2, 5, 0)
    enc[0, 7, 7] = (10, 0, 0)
    enc[0, 1, 1] = (8, 1, 0)
    o.reset()
    r = o.reset_to_encoding(enc)
    return r["image_u8"][0][..., 0].T, int(r["dir_u8"][0])


def _engines():
    from tests.hostsim import sim
    return [orc.OracleMg, sim.HostSimMg]


@pytest.mark.parametrize("cls", _engines())
def test_process_vis_open_field_sees_everything(cls):
    img, _ = _opaque_view(cls, [])
    assert (img == 1).all()                                          # empty cells everywhere, none masked to 0


@pytest.mark.parametrize("cls", _engines())
def test_process_vis_walls_on_four_sides_leak_diagonally(cls):
    """Walls N/E/S/W of the agent (symmetric, so the random start direction does not matter).  Row of the agent:
    the side walls are lit but block, so the corners stay dark; the row above is lit through the agent's own cell
    and its two diagonal neighbours are empty, which re-lights everything further up (gym-minigrid's diagonal leak)."""
    img, _ = _opaque_view(cls, [(8, 7), (6, 7), (7, 8), (7, 6)])
    expect = np.array([[1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 2, 1, 1], [0, 2, 1, 2, 0]])
    assert np.array_equal(img, expect)


@pytest.mark.parametrize("cls", _engines())
def test_process_vis_fully_enclosed_agent_sees_only_the_ring(cls):
    ring = [(x, y) for x in (6, 7, 8) for y in (6, 7, 8) if (x, y) != (7, 7)]
    img, _ = _opaque_view(cls, ring)
    expect = np.array([[0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 2, 2, 2, 0], [0, 2, 1, 2, 0]])
    assert np.array_equal(img, expect)


@pytest.mark.parametrize("cls", _engines())
def test_process_vis_wall_line_ahead_hides_what_is_behind(cls):
    """A full line of walls two cells ahead in every direction (a 5x5 box around the agent): the row of the box
    is visible, nothing beyond it."""
    box = [(x, y) for x in range(5, 10) for y in range(5, 10) if x in (5, 9) or y in (5, 9)]
    img, _ = _opaque_view(cls, box)
    expect = np.array([[0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [2, 2, 2, 2, 2], [2, 1, 1, 1, 2], [2, 1, 1, 1, 2]])
    assert np.array_equal(img, expect)


def test_edit_locations_follow_cpython_set_order():
    """mutate_level iterates `list(set(np.random.randint(...)))` (adversarial.py:327): which of two edits of the same kind
    wins depends on the set's iteration order.  The oracle's restatement of CPython's set (and with it the device code,
    compared with the oracle on every mutate call) against the interpreter's own answer, resizes included."""
    import ctypes as C
    L = orc.lib()
    rng = np.random.default_rng(0)
    for it in range(4000):
        n = int(rng.integers(1, 65))
        v = rng.integers(0, int(rng.choice([16, 64, 169])), n).astype(np.int64)
        want = [int(x) for x in list(set(v))]                      # a set of np.int64 scalars, as the reference builds it
        vi, out = v.astype(np.int32), np.zeros(64, np.int32)
        m = L.orc_py_set_order(vi.ctypes.data_as(C.c_void_p), C.c_int32(n), out.ctypes.data_as(C.c_void_p))
        assert out[:m].tolist() == want, (v.tolist(), want, out[:m].tolist())
