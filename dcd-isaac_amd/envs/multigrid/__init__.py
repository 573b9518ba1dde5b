from . import adversarial  # noqa: F401  (registers the MultiGrid-*Adversarial* ids)
from . import maze  # noqa: F401         (registers the fixed evaluation mazes)
from . import mst_maze  # noqa: F401     (PerfectMaze S/M/L/XL: a new random spanning-tree maze per episode)
from . import cluttered  # noqa: F401    (single-agent cluttered grids, procedural)
