from .multigrid_models import MultigridNetwork  # noqa: F401
from . import fused_ppo  # noqa: F401
from .car_racing_models import CarRacingNetwork, CarRacingBezierAdversaryEnvNetwork  # noqa: F401
from .walker_models import BipedalWalkerStudentPolicy, BipedalWalkerAdversaryPolicy  # noqa: F401
