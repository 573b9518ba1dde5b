"""Cluttered goal-seeking grids, single-agent variants (reference: envs/multigrid/cluttered.py:25-214).

PROCEDURAL description objects (see mst_maze.py): `_gen_grid` (cluttered.py:37-52) runs on every reset — border,
a goal (random or bottom-right), `n_clutter` walls (or lava) dropped on uniformly random EMPTY cells
(`place_obj` rejection sampling = uniform over free cells), then the agent on a random free cell facing right
(`place_agent` -> `place_agent_at_pos(rand_dir=True)`, multigrid.py:651-672).  The 3-agent id
`MultiGrid-Cluttered-v0` is outside the single-agent hot path and is not registered.
"""
import numpy as np

from envs import spaces

from . import register


class ClutteredMultiGrid:
    is_fixed_level = True
    is_procedural = True

    def __init__(self, size=15, n_agents=1, n_clutter=25, randomize_goal=True, agent_view_size=5, max_steps=250,
                 walls_are_lava=False, minigrid_mode=True, seed=None):
        if n_agents != 1:
            raise NotImplementedError("multi-agent cluttered grids are outside the single-agent MultiGrid hot path")
        self.size = self.width = self.height = size
        self.n_clutter, self.randomize_goal, self.walls_are_lava = n_clutter, randomize_goal, walls_are_lava
        self.agent_view_size, self.max_steps = agent_view_size, max_steps
        self.np_random = np.random.RandomState(seed)
        self.n_agents, self.minigrid_mode, self.see_through_walls = 1, minigrid_mode, True
        self.spec = None
        self._max_episode_steps = None
        self.action_space = spaces.Discrete(7)
        self.observation_space = spaces.Dict({
            "image": spaces.Box(0, 255, (agent_view_size, agent_view_size, 3), "uint8"),
            "direction": spaces.Box(0, 3, (1,), "uint8")})

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)

    def _free_cell(self, enc):
        xs, ys = np.nonzero(enc[..., 0] == 1)                         # empty cells
        if len(xs) == 0:
            raise RuntimeError("rejection sampling failed in place_obj")     # gym.error.RetriesExceededError
        k = self.np_random.randint(len(xs))
        return int(xs[k]), int(ys[k])

    def level_encoding(self):
        S = self.size
        enc = np.zeros((S, S, 3), np.uint8)
        enc[..., 0] = 1
        enc[0, :], enc[-1, :], enc[:, 0], enc[:, -1] = (2, 5, 0), (2, 5, 0), (2, 5, 0), (2, 5, 0)
        gx, gy = self._free_cell(enc) if self.randomize_goal else (S - 2, S - 2)
        enc[gx, gy] = (8, 1, 0)
        for _ in range(self.n_clutter):
            x, y = self._free_cell(enc)
            enc[x, y] = (9, 0, 0) if self.walls_are_lava else (2, 5, 0)
        ax, ay = self._free_cell(enc)
        enc[ax, ay] = (10, 0, 0)
        return enc

    def engine_config(self):
        return dict(size=self.size, agent_view_size=self.agent_view_size, max_steps=self.max_steps,
                    max_episode_steps=self._max_episode_steps or 0, n_clutter=0, resample_n_clutter=0,
                    choose_goal_last=0, see_through_walls=1, random_z_dim=0, editor_actions=0, fixed_environment=0,
                    goal_noise=0.0)


def _variant(name, **kw):
    return type(name, (ClutteredMultiGrid,), {"__init__": lambda self, seed=None: ClutteredMultiGrid.__init__(self, seed=seed, **kw)})


# cluttered.py:60-143
ClutteredMultiGridSingle6x6 = _variant("ClutteredMultiGridSingle6x6", size=6, n_clutter=5, max_steps=50, minigrid_mode=False)
ClutteredMultiGridSingle = _variant("ClutteredMultiGridSingle", size=15, n_clutter=25, max_steps=250, minigrid_mode=False)
Cluttered40Minigrid = _variant("Cluttered40Minigrid", n_clutter=40)
Cluttered10Minigrid = _variant("Cluttered10Minigrid", n_clutter=10)
Cluttered50Minigrid = _variant("Cluttered50Minigrid", n_clutter=50)
Cluttered5Minigrid = _variant("Cluttered5Minigrid", n_clutter=5)
Cluttered1MinigridMini = _variant("Cluttered1MinigridMini", n_clutter=1, size=6)
Cluttered6MinigridMini = _variant("Cluttered6MinigridMini", n_clutter=6, size=6)
Cluttered7MinigridMini = _variant("Cluttered7MinigridMini", n_clutter=7, size=6)
ClutteredMinigridLava = _variant("ClutteredMinigridLava", walls_are_lava=True)
ClutteredMinigridLavaMini = _variant("ClutteredMinigridLavaMini", n_clutter=4, size=6, walls_are_lava=True)
ClutteredMinigridLavaMedium = _variant("ClutteredMinigridLavaMedium", n_clutter=15, size=10, walls_are_lava=True)
Cluttered15MinigridMedium = _variant("Cluttered15MinigridMedium", n_clutter=15, size=10)

for _id, _cls in [
        ("MultiGrid-Cluttered-Single-v0", "ClutteredMultiGridSingle"),
        ("MultiGrid-Cluttered-Single-6x6-v0", "ClutteredMultiGridSingle6x6"),
        ("MultiGrid-Cluttered40-Minigrid-v0", "Cluttered40Minigrid"),
        ("MultiGrid-Cluttered10-Minigrid-v0", "Cluttered10Minigrid"),
        ("MultiGrid-Cluttered50-Minigrid-v0", "Cluttered50Minigrid"),
        ("MultiGrid-Cluttered5-Minigrid-v0", "Cluttered5Minigrid"),
        ("MultiGrid-MiniCluttered1-Minigrid-v0", "Cluttered1MinigridMini"),
        ("MultiGrid-MiniCluttered6-Minigrid-v0", "Cluttered6MinigridMini"),
        ("MultiGrid-MiniCluttered7-Minigrid-v0", "Cluttered7MinigridMini"),
        ("MultiGrid-Cluttered-Lava-Minigrid-v0", "ClutteredMinigridLava"),
        ("MultiGrid-MiniCluttered-Lava-Minigrid-v0", "ClutteredMinigridLavaMini"),
        ("MultiGrid-MediumCluttered-Lava-Minigrid-v0", "ClutteredMinigridLavaMedium"),
        ("MultiGrid-MediumCluttered15-Minigrid-v0", "Cluttered15MinigridMedium")]:
    register.register(env_id=_id, entry_point=f"{__name__}:{_cls}")
