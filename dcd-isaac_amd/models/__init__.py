from .multigrid_models import MultigridNetwork  # noqa: F401
