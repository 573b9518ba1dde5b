"""MultiGrid id registration helper (reference: envs/multigrid/register.py:24-43)."""
from envs.registration import register as _register, registry

env_list = []


def register(env_id, entry_point, reward_threshold=0.95, max_episode_steps=None):
    assert env_id.startswith("MultiGrid-")
    if env_id in registry.env_specs:
        del registry.env_specs[env_id]
    else:
        env_list.append(env_id)
    kwargs = dict(id=env_id, entry_point=entry_point, reward_threshold=reward_threshold)
    if max_episode_steps:
        kwargs["max_episode_steps"] = max_episode_steps
    _register(**kwargs)
