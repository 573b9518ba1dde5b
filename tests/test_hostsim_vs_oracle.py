"""CPU: the DEVICE env logic (dcd-isaac_amd/csrc/mg_env.h compiled for the host by tests/hostsim)
against the oracle over randomised call sequences, bit-exact.  This is the pre-GPU check of the
packed-record logic; the kernels themselves are covered by tests/test_gpu_env_parity.py."""
import numpy as np
import pytest

from oracle import orc
from tests import mg_diff
from tests.hostsim import sim

CONFIGS = [
    (0, dict(n_clutter=25, choose_goal_last=1), False),
    (1, dict(size=6, n_clutter=7, max_steps=50, max_episode_steps=50, choose_goal_last=1), False),
    (2, dict(n_clutter=25, choose_goal_last=1, max_episode_steps=50), True),
    (3, dict(resample_n_clutter=1, n_clutter=60, choose_goal_last=1, editor_actions=1), False),
    (4, dict(choose_goal_last=0, goal_noise=0.3, n_clutter=50, editor_actions=1), True),
    (5, dict(size=10, n_clutter=30, max_steps=200, max_episode_steps=200, agent_view_size=7, editor_actions=0), True),
    (6, dict(n_clutter=25, choose_goal_last=1, agent_view_size=3, fixed_environment=1), False),
    # opaque walls: views masked by Grid.process_vis (the ...OpaqueWalls... ids, multigrid.py:1000-1002)
    (7, dict(n_clutter=25, choose_goal_last=1, see_through_walls=0), False),
    (8, dict(n_clutter=50, choose_goal_last=1, see_through_walls=0, max_episode_steps=40), True),     # doors/keys/boxes + truncated obs
    (9, dict(size=10, n_clutter=30, max_steps=200, max_episode_steps=60, agent_view_size=7, see_through_walls=0), True),
    (10, dict(size=6, n_clutter=7, max_steps=50, max_episode_steps=50, agent_view_size=3, see_through_walls=0), False),
]


@pytest.mark.parametrize("seed,kw,exotic", CONFIGS)
def test_device_logic_matches_oracle(seed, kw, exotic):
    cfg = dict(num_envs=11, engine_seed=seed, env_index_base=1000 * seed)
    cfg.update(kw)
    x, y = orc.OracleMg(**cfg), sim.HostSimMg(**cfg)
    mg_diff.compare_state(x, y, "after create")
    mg_diff.drive(x, y, np.random.default_rng(seed), n_calls=30, steps_per_call=25, exotic=exotic)
