"""Lava / wall crossings and the four-rooms grid (reference: envs/multigrid/crossing.py:18-215, fourrooms.py:14-116,
registered as `MiniGrid-*` ids and used as zero-shot test environments, eval.py:208).

In the reference these two families subclass gym-minigrid's own `MiniGridEnv` (third-party, not in the reference tree)
rather than `MultiGridEnv`.  Here they run on the MultiGrid engine as PROCEDURAL fixed-level families (see
mst_maze.py): for a single agent on grids that hold only walls, lava and a goal the two base classes take the same
transitions — the agent never stands on an object it could erase (goal and lava end the episode on entry), turning,
blocked moves and the reward `1 - 0.9 * step_count / max_steps` coincide, and the observation is the same egocentric
window with the same (type, colour, state) encoding, masked by `Grid.process_vis` when `see_through_walls=False`
(crossing.py:28).  What cannot be checked is gym-minigrid itself (absent): parity unpinned, as for the whole env side.

Generators (same constructions as the reference):
  * crossing (`_gen_grid`, crossing.py:41-101): agent at (1,1) facing right, goal at (S-2,S-2); `num_crossings` rivers
    drawn from the even rows / columns, laid across the whole grid; then a random monotone path of openings from the
    agent's room to the goal's, one opening per river, positioned uniformly inside the current room's side;
  * four rooms (`_gen_grid`, fourrooms.py:40-84): a 19 x 19 grid split by a wall cross, one uniformly placed gap per
    wall segment, agent on a random free cell with a random direction, goal on a random free cell, max_steps 100.
"""
import numpy as np

from envs import spaces
from envs import registration

WALL, GOAL, LAVA = (2, 5, 0), (8, 1, 0), (9, 0, 0)


class _ProceduralEnv:
    is_fixed_level = True
    is_procedural = True

    def _setup(self, size, agent_view_size, max_steps, see_through_walls, seed):
        self.size = self.width = self.height = size
        self.agent_view_size, self.max_steps = agent_view_size, max_steps
        self.see_through_walls = see_through_walls
        self.np_random = np.random.RandomState(seed)
        self.n_agents, self.minigrid_mode = 1, True
        self.spec = None
        self._max_episode_steps = None
        self.action_space = spaces.Discrete(7)
        self.observation_space = spaces.Dict({
            "image": spaces.Box(0, 255, (agent_view_size, agent_view_size, 3), "uint8"),
            "direction": spaces.Box(0, 3, (1,), "uint8")})

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)

    def _empty(self):
        S = self.size
        enc = np.zeros((S, S, 3), np.uint8)
        enc[..., 0] = 1
        enc[0, :], enc[-1, :], enc[:, 0], enc[:, -1] = WALL, WALL, WALL, WALL
        return enc

    def engine_config(self):
        return dict(size=self.size, agent_view_size=self.agent_view_size, max_steps=self.max_steps,
                    max_episode_steps=self._max_episode_steps or 0, n_clutter=0, resample_n_clutter=0,
                    choose_goal_last=0, see_through_walls=int(bool(self.see_through_walls)), random_z_dim=0,
                    editor_actions=0, fixed_environment=0, goal_noise=0.0)


class CrossingEnv(_ProceduralEnv):
    def __init__(self, size=9, num_crossings=1, obstacle_type="lava", seed=None):
        assert size % 2 == 1, "odd size"                              # crossing.py:42
        self.num_crossings, self.obstacle = num_crossings, (LAVA if obstacle_type == "lava" else WALL)
        self._setup(size, 5, 4 * size * size, False, seed)           # crossing.py:24-31

    def level_encoding(self):
        S, rng = self.size, self.np_random
        enc = self._empty()
        enc[S - 2, S - 2] = GOAL
        rivers = [("v", i) for i in range(2, S - 2, 2)] + [("h", j) for j in range(2, S - 2, 2)]
        order = rng.permutation(len(rivers))
        rivers = [rivers[k] for k in order[:self.num_crossings]]
        rivers_v = sorted(pos for d, pos in rivers if d == "v")
        rivers_h = sorted(pos for d, pos in rivers if d == "h")
        for j in rivers_h:
            enc[1:S - 1, j] = self.obstacle                            # a river along row j
        for i in rivers_v:
            enc[i, 1:S - 1] = self.obstacle                            # a river along column i
        path = ["h"] * len(rivers_v) + ["v"] * len(rivers_h)           # crossing a v-river moves horizontally
        path = [path[k] for k in rng.permutation(len(path))]
        limits_v, limits_h = [0] + rivers_v + [S - 1], [0] + rivers_h + [S - 1]
        room_i = room_j = 0
        for d in path:                                               # openings along a monotone path to the goal's room
            if d == "h":
                i = limits_v[room_i + 1]
                j = int(rng.choice(range(limits_h[room_j] + 1, limits_h[room_j + 1])))
                room_i += 1
            else:
                i = int(rng.choice(range(limits_v[room_i] + 1, limits_v[room_i + 1])))
                j = limits_h[room_j + 1]
                room_j += 1
            enc[i, j] = (1, 0, 0)
        enc[1, 1] = (10, 0, 0)
        return enc


class FourRoomsEnv(_ProceduralEnv):
    def __init__(self, agent_pos=None, goal_pos=None, seed=None):
        self._agent_default_pos, self._goal_default_pos = agent_pos, goal_pos
        self._setup(19, 5, 100, False, seed)                          # fourrooms.py:20-26 (MiniGridEnv default: opaque walls)

    def level_encoding(self):
        S, rng = self.size, self.np_random
        enc = self._empty()
        rw = rh = S // 2
        for j in range(2):
            for i in range(2):
                xl, yt = i * rw, j * rh
                xr, yb = xl + rw, yt + rh
                if i + 1 < 2:                                        # wall to the right of the room, one gap
                    enc[xr, yt:yt + rh] = WALL
                    enc[xr, int(rng.randint(yt + 1, yb))] = (1, 0, 0)
                if j + 1 < 2:                                        # wall below the room, one gap
                    enc[xl:xl + rw, yb] = WALL
                    enc[int(rng.randint(xl + 1, xr)), yb] = (1, 0, 0)

        def free_cell():
            xs, ys = np.nonzero(enc[..., 0] == 1)
            k = int(rng.randint(len(xs)))
            return int(xs[k]), int(ys[k])
        if self._agent_default_pos is not None:
            ax, ay = self._agent_default_pos
        else:
            ax, ay = free_cell()
        enc[ax, ay] = (10, 0, int(rng.randint(0, 4)))
        gx, gy = self._goal_default_pos if self._goal_default_pos is not None else free_cell()
        enc[gx, gy] = GOAL
        return enc


def _crossing(name, **kw):
    return type(name, (CrossingEnv,), {"__init__": lambda self, seed=None: CrossingEnv.__init__(self, seed=seed, **kw)})


LavaCrossingEnv = _crossing("LavaCrossingEnv", size=9, num_crossings=1)
LavaCrossingS9N2Env = _crossing("LavaCrossingS9N2Env", size=9, num_crossings=2)
LavaCrossingS9N3Env = _crossing("LavaCrossingS9N3Env", size=9, num_crossings=3)
LavaCrossingS11N5Env = _crossing("LavaCrossingS11N5Env", size=11, num_crossings=5)
SimpleCrossingEnv = _crossing("SimpleCrossingEnv", size=9, num_crossings=1, obstacle_type="wall")
SimpleCrossingS9N2Env = _crossing("SimpleCrossingS9N2Env", size=9, num_crossings=2, obstacle_type="wall")
SimpleCrossingS9N3Env = _crossing("SimpleCrossingS9N3Env", size=9, num_crossings=3, obstacle_type="wall")
SimpleCrossingS11N5Env = _crossing("SimpleCrossingS11N5Env", size=11, num_crossings=5, obstacle_type="wall")

for _id, _cls in [("MiniGrid-LavaCrossingS9N1-v0", "LavaCrossingEnv"), ("MiniGrid-LavaCrossingS9N2-v0", "LavaCrossingS9N2Env"),
                  ("MiniGrid-LavaCrossingS9N3-v0", "LavaCrossingS9N3Env"), ("MiniGrid-LavaCrossingS11N5-v0", "LavaCrossingS11N5Env"),
                  ("MiniGrid-SimpleCrossingS9N1-v0", "SimpleCrossingEnv"), ("MiniGrid-SimpleCrossingS9N2-v0", "SimpleCrossingS9N2Env"),
                  ("MiniGrid-SimpleCrossingS9N3-v0", "SimpleCrossingS9N3Env"), ("MiniGrid-SimpleCrossingS11N5-v0", "SimpleCrossingS11N5Env"),
                  ("MiniGrid-FourRooms-v0", "FourRoomsEnv")]:
    if _id in registration.registry.env_specs:
        del registration.registry.env_specs[_id]
    registration.register(id=_id, entry_point=f"{__name__}:{_cls}")
