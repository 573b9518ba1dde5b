"""Host-side mirror of the reference's `envs` package for the MultiGrid hot path."""
from . import registration  # noqa: F401
from . import multigrid  # noqa: F401
