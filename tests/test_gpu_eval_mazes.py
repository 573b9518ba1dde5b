"""-m gpu: the fixed evaluation mazes on the same step kernel — engine vs oracle on the real maze
encodings, maze invariants, and the Evaluator end to end."""
import numpy as np
import pytest
import torch

from oracle import orc
from tests import mg_diff

pytestmark = pytest.mark.gpu
NAMES = ["MultiGrid-Maze-v0", "MultiGrid-Maze2-v0", "MultiGrid-Maze3-v0", "MultiGrid-SmallCorridor-v0", "MultiGrid-Labyrinth-v0",
         "MultiGrid-Labyrinth2-v0", "MultiGrid-SixteenRooms-v0", "MultiGrid-SixteenRoomsFewerDoors-v0",
         "MultiGrid-NineRooms-v0", "MultiGrid-NineRoomsFewerDoors-v0", "MultiGrid-MiniMaze-v0", "MultiGrid-MediumMaze-v0"]


@pytest.mark.parametrize("name", NAMES)
def test_maze_rollout_matches_oracle(name):
    from envs.registration import make
    from tests.hip_adapter import HipMg
    d = make(name)
    B = 40
    cfg = dict(num_envs=B, **d.engine_config())
    x, y = orc.OracleMg(**cfg), HipMg(**cfg)
    enc = np.stack([make(name).level_encoding() for _ in range(B)])
    ox, oy = x.reset_to_fixed_level(enc), y.reset_to_fixed_level(enc)
    mg_diff.assert_same(ox, oy, "reset")
    st = x.get_state()
    assert (st[:, 2] == 0).all() and (st[:, 0] == d.start_pos[0]).all() and (st[:, 1] == d.start_pos[1]).all()
    met = x.get_metrics()
    assert (met["passable"] == 1).all()                     # every benchmark maze is solvable
    rng = np.random.default_rng(0)
    tx = ty = None
    for t in range(2 * d.max_steps // 3):
        a = rng.choice(7, size=B, p=[0.2, 0.2, 0.5, 0.025, 0.025, 0.025, 0.025])
        ox, oy = x.step(a, 0, tx), y.step(a, 0, ty)
        tx, ty = ox["term_obs"], oy["term_obs"]
        mg_diff.assert_same(ox, oy, f"step {t}")
    mg_diff.compare_state(x, y, "end")
    assert np.array_equal(x.get_encodings()[..., 0] == 2, enc[..., 0] == 2)      # walls never change


def test_evaluator_interface():
    sys_path_ok = __import__("eval")
    from arguments import parser
    from util import create_parallel_env, make_agent
    args = parser.parse_args([])
    args.num_processes = 32
    venv, _ = create_parallel_env(args, device="cuda:0")
    agent = make_agent("agent", venv, args, "cuda:0")
    names = ["MultiGrid-SixteenRooms-v0", "MultiGrid-MiniMaze-v0"]
    ev = sys_path_ok.Evaluator(names, num_processes=32, num_episodes=40, device="cuda:0")
    assert ev.get_stats_keys() == [f"{k}:{n}" for n in names for k in ("solved_rate", "test_returns")]
    stats = ev.evaluate(agent)
    for n in names:
        assert 0.0 <= stats[f"solved_rate:{n}"] <= 1.0 and 0.0 <= stats[f"test_returns:{n}"] <= 1.0
    raw = ev.evaluate(agent, accumulator=None)
    assert all(len(raw[f"test_returns:{n}"]) == 40 for n in names)
    ev.close(); venv.close()


# ---- grids above 15 x 15 (csrc/mg_big.hip: the env logic at record pitch 32 / 64 / 128) and procedural families -------
@pytest.mark.parametrize("name,B", [("MultiGrid-LargeCorridor-v0", 33), ("MultiGrid-PerfectMazeMedium-v0", 70), ("MiniGrid-FourRooms-v0", 40), ("MiniGrid-LavaCrossingS11N5-v0", 50), ("MiniGrid-SimpleCrossingS9N3-v0", 50),
                                    ("MultiGrid-PerfectMazeLarge-v0", 20), ("MultiGrid-PerfectMazeXL-v0", 5)])
def test_large_maze_rollout_matches_oracle(name, B):
    from tests.hip_adapter import HipMg
    from tests.test_big_grids import _levels, rollout_matches
    descs, enc = _levels(name, B, seed=7)
    cfg = dict(num_envs=B, **descs[0].engine_config())
    cfg["max_episode_steps"] = 150
    x, y = orc.OracleMg(**cfg), HipMg(**cfg)
    rollout_matches(x, y, enc, 397, np.random.default_rng(B))


def test_teacher_ops_refuse_large_grids():
    import dcd_hip
    eng = dcd_hip.MultiGridEngine(4, size=21, n_clutter=0, random_z_dim=0, editor_actions=0)
    with pytest.raises(dcd_hip.DcdHipError, match="not available for grids above 15 x 15"):
        eng.reset_random()
    eng.close()


def test_evaluator_on_procedural_and_large_families():
    ev_mod = __import__("eval")
    from arguments import parser
    from util import create_parallel_env, make_agent
    args = parser.parse_args([])
    args.num_processes = 16
    venv, _ = create_parallel_env(args, device="cuda:0")
    agent = make_agent("agent", venv, args, "cuda:0")
    names = ["MultiGrid-PerfectMazeSmall-v0", "MultiGrid-PerfectMazeMedium-v0", "MultiGrid-LargeCorridor-v0",
             "MultiGrid-MiniCluttered6-Minigrid-v0", "MiniGrid-SimpleCrossingS9N2-v0", "MiniGrid-LavaCrossingS11N5-v0",
             "MiniGrid-FourRooms-v0"]
    ev = ev_mod.Evaluator(names, num_processes=16, num_episodes=20, device="cuda:0")
    assert ev.venv[names[0]].is_procedural and not ev.venv[names[2]].is_procedural
    before = ev.venv[names[3]]._fixed_levels.clone()
    stats = ev.evaluate(agent)
    for n in names:
        assert 0.0 <= stats[f"solved_rate:{n}"] <= 1.0 and 0.0 <= stats[f"test_returns:{n}"] <= 1.0
    assert not torch.equal(before, ev.venv[names[3]]._fixed_levels)    # finished episodes drew new levels
    ev.close(); venv.close()
