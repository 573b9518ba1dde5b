from .agent import ACAgent  # noqa: F401
from .ppo import PPO  # noqa: F401
from .storage import RolloutStorage  # noqa: F401
