from . import adversarial  # noqa: F401  (registers the MultiGrid-*Adversarial* ids)
from . import maze  # noqa: F401         (registers the fixed evaluation mazes)
