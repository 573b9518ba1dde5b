"""Perfect mazes from a random spanning tree (reference: envs/multigrid/mst_maze.py:17-163).

Description objects for the HIP engine, like envs/multigrid/maze.py — but PROCEDURAL: the reference draws a
new maze in `_gen_grid`, i.e. on every `reset()` (mst_maze.py:55-93, 95-116), so `level_encoding()` returns a
fresh level on each call and the vec env re-uploads a level whenever an episode ends (`is_procedural`).

Generator (same construction as the reference): cells at even interior coordinates are nodes of an h x w grid
graph, every other interior cell starts as wall; the edges are visited in a random order and an edge's wall is
opened iff its endpoints are not yet connected (Kruskal with union-find) -> a spanning tree = a perfect maze.
Agent and goal sit in two different corners, drawn per level (:41-53); the agent faces right
(`place_agent_at_pos` with its default rand_dir, multigrid.py:668-672).  The reference draws from gym's seeded
`np_random`; the stream is not reproducible here (gym 0.15.7 absent) — parity unpinned, as for every env RNG.
"""
import numpy as np

from envs import spaces

from . import register


class MSTMazeEnv:
    is_fixed_level = True        # levels are handed to the engine as encodings (dcd_mg_reset_to_fixed_level)
    is_procedural = True         # ... and a new one is generated for every episode

    def __init__(self, agent_view_size=5, minigrid_mode=True, max_steps=None, start_pos=None, goal_pos=None, size=15,
                 seed=None):
        if (size - 2) % 2 != 1:
            raise ValueError("Dimensions must be 2n+1")              # mst_maze.py:58
        self.size = self.width = self.height = size
        self.agent_view_size = agent_view_size
        self.max_steps = 2 * size * size if max_steps is None else max_steps    # :28-29
        self.np_random = np.random.RandomState(seed)
        self.n_agents, self.minigrid_mode, self.see_through_walls = 1, minigrid_mode, True
        self.spec = None
        self._max_episode_steps = None
        self.action_space = spaces.Discrete(7)
        self.observation_space = spaces.Dict({
            "image": spaces.Box(0, 255, (agent_view_size, agent_view_size, 3), "uint8"),
            "direction": spaces.Box(0, 3, (1,), "uint8")})
        self.bit_map = None
        self._sample_start_and_goal_pos()

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)

    def _sample_start_and_goal_pos(self):                            # :41-53
        s = self.size
        corners = [(1, 1), (s - 2, 1), (1, s - 2), (s - 2, s - 2)]
        a, g = self.np_random.choice(4, size=2, replace=False)
        self.start_pos, self.goal_pos = corners[int(a)], corners[int(g)]

    def _gen_maze(self):                                             # :55-93
        n = self.size - 2
        h = w = n // 2 + 1
        bit_map = np.ones((n, n), dtype=np.uint8)
        bit_map[0::2, 0::2] = 0                                      # the graph's nodes
        parent = np.arange(h * w)

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a

        ys, xs = np.divmod(np.arange(h * w), w)
        horiz = [(k, k + 1) for k in range(h * w) if xs[k] + 1 < w]
        vert = [(k, k + w) for k in range(h * w) if ys[k] + 1 < h]
        edges = horiz + vert
        for e in self.np_random.permutation(len(edges)):
            u, v = edges[e]
            ru, rv = find(u), find(v)
            if ru != rv:
                parent[ru] = rv
                bit_map[ys[u] + ys[v], xs[u] + xs[v]] = 0              # the wall between 2*(y,x) of u and of v
        self.bit_map = bit_map
        return bit_map

    def level_encoding(self):
        """A NEW level: uint8 [S,S,3] indexed [x][y][c] (border + maze walls, goal, agent facing right; :95-116)."""
        self._sample_start_and_goal_pos()
        self._gen_maze()
        S = self.size
        enc = np.zeros((S, S, 3), np.uint8)
        enc[..., 0] = 1
        enc[0, :], enc[-1, :], enc[:, 0], enc[:, -1] = (2, 5, 0), (2, 5, 0), (2, 5, 0), (2, 5, 0)
        enc[self.goal_pos[0], self.goal_pos[1]] = (8, 1, 0)
        enc[self.start_pos[0], self.start_pos[1]] = (10, 0, 0)
        ys, xs = np.nonzero(self.bit_map)                             # bit_map[y, x] -> wall at (x+1, y+1)
        enc[xs + 1, ys + 1] = (2, 5, 0)
        return enc

    def engine_config(self):
        return dict(size=self.size, agent_view_size=self.agent_view_size, max_steps=self.max_steps,
                    max_episode_steps=self._max_episode_steps or 0, n_clutter=0, resample_n_clutter=0,
                    choose_goal_last=0, see_through_walls=1, random_z_dim=0, editor_actions=0, fixed_environment=0,
                    goal_noise=0.0)


def _sized(name, size):
    return type(name, (MSTMazeEnv,), {"__init__": lambda self, seed=None: MSTMazeEnv.__init__(self, size=size, seed=seed)})


PerfectMazeSmall = _sized("PerfectMazeSmall", 11)
PerfectMazeMedium = _sized("PerfectMazeMedium", 21)
PerfectMazeLarge = _sized("PerfectMazeLarge", 51)
PerfectMazeXL = _sized("PerfectMazeXL", 101)

for _id, _cls in [("MultiGrid-PerfectMazeSmall-v0", "PerfectMazeSmall"), ("MultiGrid-PerfectMazeMedium-v0", "PerfectMazeMedium"),
                  ("MultiGrid-PerfectMazeLarge-v0", "PerfectMazeLarge"), ("MultiGrid-PerfectMazeXL-v0", "PerfectMazeXL")]:
    register.register(env_id=_id, entry_point=f"{__name__}:{_cls}")
