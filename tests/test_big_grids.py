"""CPU: grids above 15 x 15 (LargeCorridor 21, PerfectMaze 21 / 51 / 101) — the device env logic
compiled for the wider record pitches (tests/hostsim with -DDCD_PITCH=32/64/128: the same code csrc/mg_big.hip runs
on the GPU) against the oracle on real maze levels, bit-exact per step; and the level generators' invariants."""
import numpy as np
import pytest

from oracle import orc
from tests import mg_diff
from tests.hostsim import sim

CASES = [("MultiGrid-LargeCorridor-v0", 32, 6), ("MultiGrid-PerfectMazeMedium-v0", 32, 6), ("MiniGrid-FourRooms-v0", 32, 8),
         ("MultiGrid-PerfectMazeLarge-v0", 64, 3), ("MultiGrid-PerfectMazeXL-v0", 128, 2)]


def _levels(name, n, seed=0):
    import envs.multigrid  # noqa: F401
    from envs.registration import make
    descs = []
    for i in range(n):
        d = make(name)
        if getattr(d, "is_procedural", False):
            d.seed(seed * 100 + i)
        descs.append(d)
    return descs, np.stack([d.level_encoding() for d in descs])


def rollout_matches(x, y, enc, steps, rng, p_forward=0.6):
    mg_diff.assert_same(x.reset_to_fixed_level(enc), y.reset_to_fixed_level(enc), "reset")
    mg_diff.compare_state(x, y, "after reset")
    B = enc.shape[0]
    tx = ty = None
    p = [(1 - p_forward) / 2 - 0.03, (1 - p_forward) / 2 - 0.03, p_forward, 0.015, 0.015, 0.015, 0.015]
    for t in range(steps):
        a = rng.choice(7, size=B, p=p)
        ox, oy = x.step(a, 0, tx), y.step(a, 0, ty)
        tx, ty = ox["term_obs"], oy["term_obs"]
        mg_diff.assert_same(ox, oy, f"step {t}")
    mg_diff.compare_state(x, y, "end")
    st = x.get_state()
    assert st[:, 4].max() > 0 and (st[:, 0] != st[:, 8]).any()          # agents moved away from their start cells
    mg_diff.assert_same(x.reset_agent(), y.reset_agent(), "reset_agent")


@pytest.mark.parametrize("name,pitch,B", CASES)
def test_wide_pitch_logic_matches_oracle(name, pitch, B):
    descs, enc = _levels(name, B)
    cfg = dict(num_envs=B, **descs[0].engine_config())
    cfg["max_episode_steps"] = 150                                  # time limits fire inside the test (terminal obs)
    x, y = orc.OracleMg(**cfg), sim.host_sim_class(pitch)(**cfg)
    rollout_matches(x, y, enc, 397, np.random.default_rng(pitch))


def test_default_pitch_refuses_large_grids():
    with pytest.raises(Exception):
        sim.HostSimMg(num_envs=1, size=21)


@pytest.mark.parametrize("size", [11, 21, 51])
def test_mst_maze_is_a_perfect_maze(size):
    """Spanning tree: the open cells form a connected, cycle-free graph (|edges| = |cells| - 1); agent and goal sit
    in two different corners; every episode gets a new maze."""
    from envs.multigrid.mst_maze import MSTMazeEnv
    d = MSTMazeEnv(size=size, seed=size)
    encs = [d.level_encoding() for _ in range(3)]
    assert any(not np.array_equal(encs[0], e) for e in encs[1:])
    for enc in encs:
        open_ = enc[..., 0] != 2
        cells = int(open_.sum())
        edges = int((open_[1:, :] & open_[:-1, :]).sum() + (open_[:, 1:] & open_[:, :-1]).sum())
        assert edges == cells - 1
        seen = np.zeros_like(open_)
        stack = [tuple(np.argwhere(enc[..., 0] == 10)[0])]
        while stack:
            x, y = stack.pop()
            if seen[x, y] or not open_[x, y]:
                continue
            seen[x, y] = True
            stack += [(x + 1, y), (x - 1, y), (x, y + 1), (x, y - 1)]
        assert seen.sum() == cells                                    # connected: the goal is reachable
        corners = {(1, 1), (size - 2, 1), (1, size - 2), (size - 2, size - 2)}
        a, g = tuple(np.argwhere(enc[..., 0] == 10)[0]), tuple(np.argwhere(enc[..., 0] == 8)[0])
        assert a in corners and g in corners and a != g
        assert (enc[0, :, 0] == 2).all() and (enc[-1, :, 0] == 2).all() and (enc[:, 0, 0] == 2).all() and (enc[:, -1, 0] == 2).all()
    assert d.max_steps == 2 * size * size                             # mst_maze.py:28-29


def test_cluttered_levels():
    import envs.multigrid  # noqa: F401
    from envs.registration import make
    for name, n_clutter, lava, size in [("MultiGrid-Cluttered40-Minigrid-v0", 40, False, 15),
                                        ("MultiGrid-MediumCluttered-Lava-Minigrid-v0", 15, True, 10),
                                        ("MultiGrid-Cluttered-Single-6x6-v0", 5, False, 6)]:
        d = make(name)
        d.seed(1)
        enc = d.level_encoding()
        assert enc.shape == (size, size, 3)
        inner = enc[1:-1, 1:-1, 0]
        assert (inner == (9 if lava else 2)).sum() == n_clutter and (inner == 8).sum() == 1 and (inner == 10).sum() == 1
        assert not np.array_equal(enc, d.level_encoding())


def _reachable(enc):
    free = ~np.isin(enc[..., 0], (2, 9))                              # not wall, not lava
    seen = np.zeros_like(free)
    stack = [tuple(np.argwhere(enc[..., 0] == 10)[0])]
    while stack:
        x, y = stack.pop()
        if seen[x, y] or not free[x, y]:
            continue
        seen[x, y] = True
        stack += [(x + 1, y), (x - 1, y), (x, y + 1), (x, y - 1)]
    gx, gy = np.argwhere(enc[..., 0] == 8)[0]
    return bool(seen[gx, gy])


@pytest.mark.parametrize("name,size,n,obst", [("MiniGrid-LavaCrossingS9N1-v0", 9, 1, 9), ("MiniGrid-LavaCrossingS11N5-v0", 11, 5, 9),
                                              ("MiniGrid-SimpleCrossingS9N3-v0", 9, 3, 2), ("MiniGrid-SimpleCrossingS11N5-v0", 11, 5, 2)])
def test_crossing_levels(name, size, n, obst):
    """crossing.py:41-101: `n` full rivers on even rows / columns, one opening per river along a monotone path, so the
    goal in the far corner is always reachable from the agent in (1,1) without touching an obstacle."""
    import envs.multigrid  # noqa: F401
    from envs.registration import make
    d = make(name)
    d.seed(size * 10 + n)
    assert d.size == size and d.max_steps == 4 * size * size and not d.see_through_walls and d.agent_view_size == 5
    levels = [d.level_encoding() for _ in range(20)]
    assert len({l.tobytes() for l in levels}) > 5
    for enc in levels:
        assert tuple(enc[1, 1]) == (10, 0, 0) and tuple(enc[size - 2, size - 2]) == (8, 1, 0)
        inner = enc[1:-1, 1:-1, 0]
        rows = [j for j in range(size - 2) if (inner[:, j] == obst).sum() >= size - 3]      # a river minus its one opening
        cols = [i for i in range(size - 2) if (inner[i, :] == obst).sum() >= size - 3]
        assert len(rows) + len(cols) == n and all((k + 1) % 2 == 0 for k in rows + cols)
        assert _reachable(enc)


def test_four_rooms_levels():
    import envs.multigrid  # noqa: F401
    from envs.registration import make
    d = make("MiniGrid-FourRooms-v0")
    d.seed(4)
    assert d.size == 19 and d.max_steps == 100
    dirs = set()
    for _ in range(20):
        enc = d.level_encoding()
        t = enc[..., 0]
        assert (t[9, 1:18] == 2).sum() == 15 and (t[1:18, 9] == 2).sum() == 15     # a wall cross with 2 + 2 gaps
        assert (t == 8).sum() == 1 and (t == 10).sum() == 1 and _reachable(enc)
        dirs.add(int(enc[..., 2][t == 10][0]))
    assert len(dirs) > 1                                              # random start direction (fourrooms.py:72-76)
