"""Adversarial MultiGrid level families and their registered ids
(reference: envs/multigrid/adversarial.py:59-149 constructor, :584-671 variants, :680-785 ids).

An instance is a DESCRIPTION of the env family — constructor arguments, spaces, TimeLimit — that
`envs.wrappers.HipAdversarialVecEnv` turns into a `DcdMgConfig` for the GPU engine; the
transition functions of the reference class (reset / step / step_adversary / reset_random /
reset_to_level / mutate_level, adversarial.py:194-581, multigrid.py:943-1055) live in
dcd-isaac_amd/csrc/mg_env.h and run batched.  Per-env stepping of a single Python object is not
part of the hot path and is not provided.
"""
from envs import spaces
from . import register

EDITOR_ACTION_SPACES = {
    "walls_none": {0: "-", 1: "."},
    "walls_none_goal": {0: "-", 1: ".", 2: "g"},
    "walls_none_agent_goal": {0: "-", 1: ".", 2: "a", 3: "g"},
}
_EDITOR_CODE = {"walls_none": 0, "walls_none_goal": 1, "walls_none_agent_goal": 2}


class AdversarialEnv:
    def __init__(self, n_clutter=50, resample_n_clutter=False, size=15, agent_view_size=5, max_steps=250,
                 goal_noise=0.0, random_z_dim=50, choose_goal_last=False, see_through_walls=True, seed=0,
                 editor_actions="walls_none_agent_goal", fixed_environment=False):
        self.n_clutter = n_clutter
        self.resample_n_clutter = resample_n_clutter
        self.width = self.height = self.size = size
        self.agent_view_size = agent_view_size
        self.max_steps = max_steps
        self.goal_noise = goal_noise
        self.random_z_dim = random_z_dim
        self.choose_goal_last = choose_goal_last
        self.see_through_walls = see_through_walls
        self.seed_value = seed
        self.editor_actions_name = editor_actions
        self.editor_actions = list(EDITOR_ACTION_SPACES[editor_actions].values())
        self.fixed_environment = fixed_environment
        self.n_agents = 1
        self.minigrid_mode = True
        self.adversary_max_steps = n_clutter + 2
        self.spec = None
        self._max_episode_steps = None
        # spaces (multigrid.py:407-440, adversarial.py:127-145)
        self.action_space = spaces.Discrete(7)
        self.observation_space = spaces.Dict({
            "image": spaces.Box(0, 255, (agent_view_size, agent_view_size, 3), "uint8"),
            "direction": spaces.Box(0, 3, (1,), "uint8")})
        self.adversary_action_dim = (size - 2) ** 2
        self.adversary_action_space = spaces.Discrete(self.adversary_action_dim)
        self.adversary_observation_space = spaces.Dict({
            "image": spaces.Box(0, 255, (size, size, 3), "uint8"),
            "time_step": spaces.Box(0, self.adversary_max_steps, (1,), "uint8"),
            "random_z": spaces.Box(0, 1.0, (random_z_dim,), "float32")})

    @property
    def processed_action_dim(self):
        return 1

    def engine_config(self):
        """Keyword arguments of dcd_hip.MultiGridEngine / DcdMgConfig (include/dcd_hip.h)."""
        return dict(size=self.size, agent_view_size=self.agent_view_size, max_steps=self.max_steps,
                    max_episode_steps=self._max_episode_steps or 0, n_clutter=self.n_clutter,
                    resample_n_clutter=int(bool(self.resample_n_clutter)),
                    choose_goal_last=int(bool(self.choose_goal_last)),
                    see_through_walls=int(bool(self.see_through_walls)), random_z_dim=self.random_z_dim,
                    editor_actions=_EDITOR_CODE[self.editor_actions_name],
                    fixed_environment=int(bool(self.fixed_environment)), goal_noise=float(self.goal_noise))


def _family(name, **fixed):
    """Variant whose constructor only takes (fixed_environment, seed), as in the reference (:596-660)."""
    def __init__(self, fixed_environment=False, seed=None):
        AdversarialEnv.__init__(self, fixed_environment=fixed_environment, seed=seed, **fixed)
    return type(name, (AdversarialEnv,), {"__init__": __init__})


class MiniAdversarialEnv(AdversarialEnv):
    def __init__(self):
        super().__init__(n_clutter=7, size=6, agent_view_size=5, max_steps=50)


class NoisyAdversarialEnv(AdversarialEnv):
    def __init__(self):
        super().__init__(goal_noise=0.3)


class MediumAdversarialEnv(AdversarialEnv):
    def __init__(self):
        super().__init__(n_clutter=30, size=10, agent_view_size=5, max_steps=200)


GoalLastAdversarialEnv = _family("GoalLastAdversarialEnv", choose_goal_last=True, max_steps=250)
GoalLastAdversarialEnv30 = _family("GoalLastAdversarialEnv30", choose_goal_last=True, n_clutter=30, max_steps=250)
GoalLastAdversarialEnv60 = _family("GoalLastAdversarialEnv60", choose_goal_last=True, n_clutter=60, max_steps=250)
GoalLastOpaqueWallsAdversarialEnv = _family("GoalLastOpaqueWallsAdversarialEnv", choose_goal_last=True,
                                            see_through_walls=False, max_steps=250)
GoalLastFewerBlocksAdversarialEnv = _family("GoalLastFewerBlocksAdversarialEnv", choose_goal_last=True,
                                            n_clutter=25, max_steps=250)
GoalLastFewerBlocksAdversarialEnv_WN = _family("GoalLastFewerBlocksAdversarialEnv_WN", choose_goal_last=True,
                                               n_clutter=25, max_steps=250, editor_actions="walls_none")
GoalLastFewerBlocksAdversarialEnv_WNG = _family("GoalLastFewerBlocksAdversarialEnv_WNG", choose_goal_last=True,
                                                n_clutter=25, max_steps=250, editor_actions="walls_none_goal")
GoalLastVariableBlocksAdversarialEnv_WNG = _family("GoalLastVariableBlocksAdversarialEnv_WNG", choose_goal_last=True,
                                                   n_clutter=60, resample_n_clutter=True, max_steps=250,
                                                   editor_actions="walls_none_goal")
GoalLastVariableBlocksAdversarialEnv = _family("GoalLastVariableBlocksAdversarialEnv", choose_goal_last=True,
                                               n_clutter=60, resample_n_clutter=True, max_steps=250)
GoalLastEmptyAdversarialEnv_WNG = _family("GoalLastEmptyAdversarialEnv_WNG", choose_goal_last=True, n_clutter=0,
                                          max_steps=250, editor_actions="walls_none_goal")
GoalLastFewerBlocksOpaqueWallsAdversarialEnv = _family("GoalLastFewerBlocksOpaqueWallsAdversarialEnv",
                                                       choose_goal_last=True, n_clutter=25, see_through_walls=False,
                                                       max_steps=250)


class MiniGoalLastAdversarialEnv(AdversarialEnv):
    def __init__(self, fixed_environment=False, seed=None):
        super().__init__(n_clutter=7, size=6, agent_view_size=5, max_steps=50, choose_goal_last=True,
                         fixed_environment=fixed_environment, seed=seed)


class FixedAdversarialEnv(AdversarialEnv):
    def __init__(self):
        super().__init__(n_clutter=50, size=15, agent_view_size=5, max_steps=50, fixed_environment=True)


class EmptyMiniFixedAdversarialEnv(AdversarialEnv):
    def __init__(self):
        super().__init__(n_clutter=0, size=6, agent_view_size=5, max_steps=50, fixed_environment=True)


_M = __name__
for _id, _cls, _steps in [
        ("MultiGrid-Adversarial-v0", "AdversarialEnv", 250),
        ("MultiGrid-MiniAdversarial-v0", "MiniAdversarialEnv", 50),
        ("MultiGrid-NoisyAdversarial-v0", "NoisyAdversarialEnv", 250),
        ("MultiGrid-MediumAdversarial-v0", "MediumAdversarialEnv", 200),
        ("MultiGrid-GoalLastAdversarial-v0", "GoalLastAdversarialEnv", 250),
        ("MultiGrid-GoalLastOpaqueWallsAdversarial-v0", "GoalLastOpaqueWallsAdversarialEnv", 250),
        ("MultiGrid-GoalLastFewerBlocksAdversarial-v0", "GoalLastFewerBlocksAdversarialEnv", 250),
        ("MultiGrid-GoalLastFewerBlocksAdversarial-EditWN-v0", "GoalLastFewerBlocksAdversarialEnv_WN", 250),
        ("MultiGrid-GoalLastFewerBlocksAdversarial-EditWNG-v0", "GoalLastFewerBlocksAdversarialEnv_WNG", 250),
        ("MultiGrid-GoalLastVariableBlocksAdversarialEnv-v0", "GoalLastVariableBlocksAdversarialEnv", 250),
        ("MultiGrid-GoalLastVariableBlocksAdversarialEnv-Edit-v0", "GoalLastVariableBlocksAdversarialEnv_WNG", 250),
        ("MultiGrid-GoalLastEmptyAdversarialEnv-Edit-v0", "GoalLastEmptyAdversarialEnv_WNG", 250),
        ("MultiGrid-GoalLastFewerBlocksOpaqueWallsAdversarial-v0", "GoalLastFewerBlocksOpaqueWallsAdversarialEnv", 250),
        ("MultiGrid-MiniGoalLastAdversarial-v0", "MiniGoalLastAdversarialEnv", 50),
        ("MultiGrid-FixedAdversarial-v0", "FixedAdversarialEnv", 50),
        ("MultiGrid-EmptyMiniFixedAdversarial-v0", "EmptyMiniFixedAdversarialEnv", 50),
        ("MultiGrid-GoalLastAdversarialEnv30-v0", "GoalLastAdversarialEnv30", 50),     # sic: 50 in the reference (:775-779)
        ("MultiGrid-GoalLastAdversarialEnv60-v0", "GoalLastAdversarialEnv60", 50),
]:
    register.register(env_id=_id, entry_point=f"{_M}:{_cls}", max_episode_steps=_steps)
