"""CPU: the repo's env registry and level tables against the constants the reference holds in source
(tests/golden/mg_registry.json, extracted without importing anything by tests/golden/extract_registry.py from
envs/multigrid/{adversarial,maze,mst_maze,cluttered,crossing,fourrooms}.py)."""
import json
import os

import numpy as np
import pytest

import envs  # noqa: F401  (registers every id)
from envs import registration
from envs.multigrid import adversarial as adv_mod
from envs.multigrid import maze as maze_mod

REF = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mg_registry.json")))
# the one registered id that is not served: three agents stepping in a shuffled order (multigrid.py:952-958) are
# outside the single-agent hot path
NOT_SERVED = {"MultiGrid-Cluttered-v0"}


def _regs():
    return [(fname, r) for fname, mod in sorted(REF.items()) for r in mod["registrations"]]


def _resolved_kwargs(fname, cls, root):
    """Constructor arguments of `root` that `cls()` ends up with: root's defaults overridden along the class chain."""
    classes = REF[fname]["classes"]
    chain = [cls]
    while chain[-1] != root:
        chain.append(classes[chain[-1]]["bases"][0])
    kw = dict(classes[root]["init_defaults"])
    for c in reversed(chain[:-1]):
        for k, v in classes[c]["super_kwargs"].items():
            if isinstance(v, dict) and "name" in v:                    # forwarded constructor argument: its default
                d = classes[c]["init_defaults"].get(v["name"], v)
                v = d
            kw[k] = v
    return kw


def test_every_reference_id_is_registered_with_the_same_class_and_time_limit():
    seen = set()
    for fname, r in _regs():
        seen.add(r["id"])
        if r["id"] in NOT_SERVED:
            assert r["id"] not in registration.registry.env_specs
            continue
        spec = registration.spec(r["id"])
        assert spec.entry_point.split(":")[-1] == r["cls"], r["id"]
        assert spec.max_episode_steps == r["max_episode_steps"], r["id"]
    assert len(seen) == 58
    ours = {i for i in registration.registry.env_specs if i.startswith(("MultiGrid-", "MiniGrid-"))}
    assert ours == seen - NOT_SERVED


def test_editor_action_spaces():
    ref = {k: {int(i): v for i, v in d.items()} for k, d in REF["adversarial.py"]["EDITOR_ACTION_SPACES"].items()}
    assert adv_mod.EDITOR_ACTION_SPACES == ref


@pytest.mark.parametrize("r", [r for f, r in _regs() if f == "adversarial.py"], ids=lambda r: r["id"])
def test_adversarial_families(r):
    kw = _resolved_kwargs("adversarial.py", r["cls"], "AdversarialEnv")
    env = registration.make(r["id"])
    assert env._max_episode_steps == r["max_episode_steps"]
    for key in ("n_clutter", "resample_n_clutter", "size", "agent_view_size", "max_steps", "goal_noise", "random_z_dim",
                "choose_goal_last", "see_through_walls", "fixed_environment"):
        assert getattr(env, key) == kw[key], (r["id"], key)
    assert env.editor_actions_name == kw["editor_actions"]
    assert env.adversary_max_steps == kw["n_clutter"] + 2                 # adversarial.py:107


@pytest.mark.parametrize("r", [r for f, r in _regs() if f == "maze.py"], ids=lambda r: r["id"])
def test_fixed_mazes(r):
    c = REF["maze.py"]["classes"][r["cls"]]
    base = REF["maze.py"]["classes"]["MazeEnv"]
    env = registration.make(r["id"])
    if r["cls"] == "MazeEnv":
        bit_map, size = base["default_bit_map"], base["init_defaults"]["size"]
        d = base["defaults"]
        start, goal = [d["default_agent_start_x"], d["default_agent_start_y"]], [d["default_goal_start_x"], d["default_goal_start_y"]]
    else:
        a = c["assigns"]
        bit_map, size, start, goal = a["bit_map"], c["super_kwargs"]["size"], a["start_pos"], a["goal_pos"]
    assert env.size == size and env.max_steps == 2 * size * size          # maze.py:39-40
    assert np.array_equal(env.bit_map, np.array(bit_map))
    assert list(env.start_pos) == start
    if isinstance(goal[0], dict):                                          # corridor mazes: goal = (choice(cols), choice(rows))
        cols, rows = c["assigns"][goal[0]["name"]]["choice"], c["assigns"][goal[1]["name"]]["choice"]
        assert maze_mod._RANDOM_GOALS[r["cls"]] == (cols, rows)
        assert env.goal_pos[0] in cols and env.goal_pos[1] in rows
    else:
        assert list(env.goal_pos) == goal
    # the encoding the engine loads: walls from the bit map (x+1, y+1), goal, agent facing right
    enc = env.level_encoding()
    bm = np.array(bit_map)
    for y in range(size - 2):
        for x in range(size - 2):
            if bm[y, x]:
                assert tuple(enc[x + 1, y + 1]) == (2, 5, 0)
    assert tuple(enc[env.start_pos[0], env.start_pos[1]]) == (10, 0, 0) or bm[env.start_pos[1] - 1, env.start_pos[0] - 1]


@pytest.mark.parametrize("r", [r for f, r in _regs() if f == "mst_maze.py"], ids=lambda r: r["id"])
def test_perfect_mazes(r):
    size = REF["mst_maze.py"]["classes"][r["cls"]]["super_kwargs"]["size"]
    env = registration.make(r["id"])
    assert env.size == size and env.max_steps == 2 * size * size          # mst_maze.py:28-29


@pytest.mark.parametrize("r", [r for f, r in _regs() if f == "cluttered.py" and r["id"] not in NOT_SERVED], ids=lambda r: r["id"])
def test_cluttered_families(r):
    kw = _resolved_kwargs("cluttered.py", r["cls"], "ClutteredMultiGrid")
    env = registration.make(r["id"])
    assert kw["n_agents"] == 1
    for key in ("size", "n_clutter", "randomize_goal", "agent_view_size", "max_steps", "walls_are_lava"):
        assert getattr(env, key) == kw[key], (r["id"], key)


@pytest.mark.parametrize("r", [r for f, r in _regs() if f in ("crossing.py", "fourrooms.py")], ids=lambda r: r["id"])
def test_crossings_and_four_rooms(r):
    env = registration.make(r["id"])
    if r["cls"] == "FourRoomsEnv":
        kw = REF["fourrooms.py"]["classes"]["FourRoomsEnv"]["super_kwargs"]
        assert (env.size, env.max_steps, env.agent_view_size) == (kw["grid_size"], kw["max_steps"], kw["agent_view_size"])
        return
    kw = _resolved_kwargs("crossing.py", r["cls"], "CrossingEnv")
    size = kw["size"]
    assert env.size == size and env.num_crossings == kw["num_crossings"]
    assert env.max_steps == 4 * size * size and not env.see_through_walls  # crossing.py:24-31
    from envs.multigrid import crossing
    assert tuple(env.obstacle) == tuple(crossing.LAVA if kw["obstacle_type"]["name"] == "Lava" else crossing.WALL)
