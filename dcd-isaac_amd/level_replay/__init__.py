from .level_sampler import LevelSampler  # noqa: F401
from .level_store import LevelStore  # noqa: F401
