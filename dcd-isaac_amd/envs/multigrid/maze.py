"""Zero-shot evaluation mazes: fixed levels given as wall bit maps (reference: envs/multigrid/maze.py:23-94 class
MazeEnv and its variants :97-403, ids :406-469).  SURVEY §8f row 2 ("next": the callers on the evaluation side).

As for the adversarial families, an instance is a DESCRIPTION: `level_encoding()` renders the bit map, goal and
start cell into the uint8 [S,S,3] encoding that dcd_mg_reset_to_fixed_level loads; stepping is the same GPU
kernel.  The agent starts facing right (place_agent_at_pos with rand_dir=True sets dir 0, multigrid.py:668-672),
`max_steps = 2*size*size` (maze.py:39-40) and there is no TimeLimit wrapper (the ids carry no
max_episode_steps).  The bit maps are the benchmark's level DATA (rows top to bottom, '1' = wall).
Grids larger than 15x15 (LargeCorridor, 21x21) exceed the engine's record layout and raise at construction.
"""
import numpy as np

from envs import spaces
from . import register

_MAZES = {
    "MazeEnv": dict(size=15, start=(7, 1), goal=(7, 13), rows=(
        "0000010000100",
        "0111011110110",
        "0100000000000",
        "0111111110111",
        "0000000010000",
        "1111110111110",
        "0000100100000",
        "0110001101111",
        "0010100100010",
        "1010110111010",
        "1010010001000",
        "1011011101110",
        "0001000101000",
    )),
    "MiniMazeEnv": dict(size=6, start=(1, 1), goal=(1, 3), rows=(
        "0000",
        "1110",
        "0100",
        "0001",
    )),
    "MediumMazeEnv": dict(size=10, start=(5, 1), goal=(3, 8), rows=(
        "01000110",
        "01010100",
        "01011110",
        "00000100",
        "11110101",
        "00000000",
        "01111110",
        "00010000",
    )),
    "HorizontalMazeEnv": dict(size=15, start=(1, 7), goal=(13, 5), rows=(
        "0001010000100",
        "0101011110001",
        "0100000000100",
        "0111111110111",
        "0001001010100",
        "1101011010100",
        "0101000010110",
        "0101101110010",
        "0100100111010",
        "0110110101010",
        "0100010101010",
        "0101010101010",
        "0001000100000",
    )),
    "Maze3Env": dict(size=15, start=(4, 1), goal=(13, 7), rows=(
        "0000101000000",
        "0111101011110",
        "0100001010000",
        "0001111010101",
        "1101000010100",
        "0001011010110",
        "0101010010010",
        "0101010111011",
        "0100010101000",
        "0111010101110",
        "0100010100010",
        "0101110101010",
        "0100010001000",
    )),
    "SmallCorridorEnv": dict(size=15, start=(1, 7), goal=None, rows=(
        "0000000000000",
        "0101010101010",
        "0101010101010",
        "0101010101010",
        "0101010101010",
        "0101010101010",
        "0111111111110",
        "0101010101010",
        "0101010101010",
        "0101010101010",
        "0101010101010",
        "0101010101010",
        "0000000000000",
    )),
    "LargeCorridorEnv": dict(size=21, start=(1, 10), goal=None, rows=(
        "0000000000000000000",
        "0101010101010101010",
        "0101010101010101010",
        "0101010101010101010",
        "0101010101010101010",
        "0101010101010101010",
        "0101010101010101010",
        "0101010101010101010",
        "0101010101010101010",
        "0111111111111111110",
        "0101010101010101010",
        "0101010101010101010",
        "0101010101010101010",
        "0101010101010101010",
        "0101010101010101010",
        "0101010101010101010",
        "0101010101010101010",
        "0101010101010101010",
        "0000000000000000000",
    )),
    "LabyrinthEnv": dict(size=15, start=(1, 13), goal=(7, 7), rows=(
        "0000000000000",
        "0111111111110",
        "0100000000010",
        "0101111111010",
        "0101000001010",
        "0101011101010",
        "0101010101010",
        "0101010101010",
        "0100010001010",
        "0111111111010",
        "0000010000010",
        "1111010111110",
        "0000010000000",
    )),
    "Labyrinth2Env": dict(size=15, start=(1, 1), goal=(7, 7), rows=(
        "0100000000000",
        "0101111111110",
        "0101000000010",
        "0101011111010",
        "0101010001010",
        "0001010101010",
        "1111010101010",
        "0001011101010",
        "0101000001010",
        "0101111111010",
        "0100000000010",
        "0111111111110",
        "0000000000000",
    )),
    "SixteenRoomsEnv": dict(size=15, start=(2, 2), goal=(12, 12), rows=(
        "0001001001000",
        "0000000001000",
        "0001001000000",
        "1011101101110",
        "0001000000000",
        "0000001001000",
        "1101011011101",
        "0001000001000",
        "0001001000000",
        "0111101101011",
        "0001001001000",
        "0000001000000",
        "0001000001000",
    )),
    "SixteenRoomsFewerDoorsEnv": dict(size=15, start=(2, 2), goal=(12, 12), rows=(
        "0001000001000",
        "0000001001000",
        "0001001001000",
        "1111011011101",
        "0001001000000",
        "0000001001000",
        "1011111011111",
        "0001001001000",
        "0001000000000",
        "1101101101111",
        "0001001001000",
        "0000001000000",
        "0001001001000",
    )),
    "NineRoomsEnv": dict(size=15, start=(2, 2), goal=(12, 12), rows=(
        "0000000010000",
        "0000100000000",
        "0000100010000",
        "0000100010000",
        "1011101111110",
        "0000100000000",
        "0000000010000",
        "0000100010000",
        "1110111011011",
        "0000100010000",
        "0000000010000",
        "0000100000000",
        "0000100010000",
    )),
    "NineRoomsFewerDoorsEnv": dict(size=15, start=(2, 2), goal=(12, 12), rows=(
        "0000100010000",
        "0000100000000",
        "0000100010000",
        "0000100010000",
        "1011101111110",
        "0000100010000",
        "0000000010000",
        "0000100010000",
        "1110110111011",
        "0000100010000",
        "0000000010000",
        "0000100010000",
        "0000100010000",
    )),
}

# corridor variants draw the goal cell per env instance (maze.py:153-156, 180-183)
_RANDOM_GOALS = {"SmallCorridorEnv": ([3, 5, 7, 9, 11], [6, 8]), "LargeCorridorEnv": ([3, 5, 7, 9, 11, 13, 15, 17], [9, 11])}


class MazeEnv:
    """Single-agent maze specified via a bit map (description object; see module docstring)."""
    maze_name = "MazeEnv"

    def __init__(self, agent_view_size=5, minigrid_mode=True, max_steps=None, bit_map=None, start_pos=None,
                 goal_pos=None, size=None, seed=None, fixed_environment=False):
        spec = _MAZES[self.maze_name]
        size = spec["size"] if size is None else size
        if bit_map is None:
            bit_map = np.array([[int(c) for c in row] for row in spec["rows"]], dtype=np.uint8)
        bit_map = np.asarray(bit_map)
        if bit_map.shape != (size - 2, size - 2):
            raise ValueError(f"bit map {bit_map.shape} does not match size {size}")
        if size > 127:
            raise NotImplementedError(f"{type(self).__name__}: {size}x{size} grids exceed the engine's widest record (127)")
        self.bit_map = bit_map
        self.size = self.width = self.height = size
        self.agent_view_size = agent_view_size
        self.max_steps = 2 * size * size if max_steps is None else max_steps
        self.start_pos = tuple(spec["start"]) if start_pos is None else tuple(start_pos)
        if goal_pos is None:
            if self.maze_name in _RANDOM_GOALS:
                cols, rws = _RANDOM_GOALS[self.maze_name]
                goal_pos = (int(np.random.choice(cols)), int(np.random.choice(rws)))
            else:
                goal_pos = spec["goal"]
        self.goal_pos = tuple(goal_pos)
        self.n_agents, self.minigrid_mode, self.see_through_walls = 1, minigrid_mode, True
        self.seed_value = seed
        self.is_procedural = False
        self.spec = None
        self._max_episode_steps = None
        self.action_space = spaces.Discrete(7)
        self.observation_space = spaces.Dict({
            "image": spaces.Box(0, 255, (agent_view_size, agent_view_size, 3), "uint8"),
            "direction": spaces.Box(0, 3, (1,), "uint8")})

    is_fixed_level = True

    def level_encoding(self):
        """uint8 [S,S,3] indexed [x][y][c]: border + bit-map walls (grey), goal, agent facing right (_gen_grid :76-94)."""
        S = self.size
        enc = np.zeros((S, S, 3), np.uint8)
        enc[..., 0] = 1
        enc[0, :], enc[-1, :], enc[:, 0], enc[:, -1] = (2, 5, 0), (2, 5, 0), (2, 5, 0), (2, 5, 0)
        enc[self.goal_pos[0], self.goal_pos[1]] = (8, 1, 0)
        enc[self.start_pos[0], self.start_pos[1]] = (10, 0, 0)
        ys, xs = np.nonzero(self.bit_map)                 # bit_map[y, x] -> wall at (x+1, y+1), overwriting
        enc[xs + 1, ys + 1] = (2, 5, 0)
        return enc

    def engine_config(self):
        return dict(size=self.size, agent_view_size=self.agent_view_size, max_steps=self.max_steps,
                    max_episode_steps=self._max_episode_steps or 0, n_clutter=0, resample_n_clutter=0,
                    choose_goal_last=0, see_through_walls=1, random_z_dim=0, editor_actions=0, fixed_environment=0,
                    goal_noise=0.0)


def _variant(name):
    return type(name, (MazeEnv,), {"maze_name": name})

MiniMazeEnv = _variant("MiniMazeEnv")
MediumMazeEnv = _variant("MediumMazeEnv")
HorizontalMazeEnv = _variant("HorizontalMazeEnv")
Maze3Env = _variant("Maze3Env")
SmallCorridorEnv = _variant("SmallCorridorEnv")
LargeCorridorEnv = _variant("LargeCorridorEnv")
LabyrinthEnv = _variant("LabyrinthEnv")
Labyrinth2Env = _variant("Labyrinth2Env")
SixteenRoomsEnv = _variant("SixteenRoomsEnv")
SixteenRoomsFewerDoorsEnv = _variant("SixteenRoomsFewerDoorsEnv")
NineRoomsEnv = _variant("NineRoomsEnv")
NineRoomsFewerDoorsEnv = _variant("NineRoomsFewerDoorsEnv")

for _id, _cls in [
        ("MultiGrid-Maze-v0", "MazeEnv"),
        ("MultiGrid-MiniMaze-v0", "MiniMazeEnv"),
        ("MultiGrid-MediumMaze-v0", "MediumMazeEnv"),
        ("MultiGrid-Maze2-v0", "HorizontalMazeEnv"),
        ("MultiGrid-Maze3-v0", "Maze3Env"),
        ("MultiGrid-SmallCorridor-v0", "SmallCorridorEnv"),
        ("MultiGrid-LargeCorridor-v0", "LargeCorridorEnv"),
        ("MultiGrid-Labyrinth-v0", "LabyrinthEnv"),
        ("MultiGrid-Labyrinth2-v0", "Labyrinth2Env"),
        ("MultiGrid-SixteenRooms-v0", "SixteenRoomsEnv"),
        ("MultiGrid-SixteenRoomsFewerDoors-v0", "SixteenRoomsFewerDoorsEnv"),
        ("MultiGrid-NineRooms-v0", "NineRoomsEnv"),
        ("MultiGrid-NineRoomsFewerDoors-v0", "NineRoomsFewerDoorsEnv"),
]:
    register.register(env_id=_id, entry_point=f"{__name__}:{_cls}")
