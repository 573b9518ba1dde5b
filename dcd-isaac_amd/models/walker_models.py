"""BipedalWalker student policy and level-builder policy (reference: models/walker_models.py:113-167
BipedalWalkerStudentPolicy, :170-257 BipedalWalkerAdversaryPolicy, :345-376 MLPBase, :49-67 DiagGaussian; BASELINE config 4,
SURVEY §8f-4).  PyTorch-ROCm like the MultiGrid nets; parameter names and
initialisers follow the reference (same seed -> same weights, its checkpoints load): two tanh MLP towers of width 64 for
actor and critic (orthogonal, gain sqrt 2), a linear value head, a diagonal Gaussian with a state-independent log-std.
The feed-forward configuration every shipped BipedalWalker config uses; the optional GRU of the reference's base is the
masked recurrence of models/common.RNN and is not duplicated here.  The Box2D dynamics the policy acts in have no oracle
in this environment (un-vendored third-party C++; parity unpinned)."""
import numpy as np
import torch
import torch.nn as nn

from .common import DeviceAwareModule


def _ortho(m, gain=1.0):
    nn.init.orthogonal_(m.weight.data, gain=gain)
    nn.init.constant_(m.bias.data, 0)
    return m


class AddBias(nn.Module):
    def __init__(self, bias):
        super().__init__()
        self._bias = nn.Parameter(bias.unsqueeze(1))

    def forward(self, x):
        return x + (self._bias.t().view(1, -1) if x.dim() == 2 else self._bias.t().view(1, -1, 1, 1))


class FixedNormal(torch.distributions.Normal):
    def log_probs(self, actions):
        return super().log_prob(actions).sum(-1, keepdim=True)

    def entropy(self):
        return super().entropy().sum(-1)

    def mode(self):
        return self.mean


class DiagGaussian(DeviceAwareModule):
    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        self.fc_mean = _ortho(nn.Linear(num_inputs, num_outputs))
        self.logstd = AddBias(torch.zeros(num_outputs))

    def forward(self, x):
        mean = self.fc_mean(x)
        return FixedNormal(mean, self.logstd(torch.zeros_like(mean)).exp(), validate_args=False)


class MLPBase(nn.Module):
    def __init__(self, num_inputs, recurrent=False, hidden_size=64):
        super().__init__()
        if recurrent:
            raise NotImplementedError("recurrent BipedalWalker base: off in every shipped config")
        self._hidden_size = hidden_size
        tower = lambda: nn.Sequential(_ortho(nn.Linear(num_inputs, hidden_size), np.sqrt(2)), nn.Tanh(),
                                      _ortho(nn.Linear(hidden_size, hidden_size), np.sqrt(2)), nn.Tanh())
        self.actor, self.critic = tower(), tower()
        self.critic_linear = _ortho(nn.Linear(hidden_size, 1), np.sqrt(2))
        self.train()

    is_recurrent = False
    recurrent_hidden_state_size = 1

    @property
    def output_size(self):
        return self._hidden_size

    def forward(self, inputs, rnn_hxs, masks):
        return self.critic_linear(self.critic(inputs)), self.actor(inputs), rnn_hxs


class BipedalWalkerStudentPolicy(DeviceAwareModule):
    def __init__(self, obs_shape, action_space, recurrent=False, base_kwargs=None):
        super().__init__()
        self.base = MLPBase(obs_shape[0], recurrent=recurrent, **(base_kwargs or {}))
        self.dist = DiagGaussian(self.base.output_size, action_space.shape[0])

    @property
    def is_recurrent(self):
        return self.base.is_recurrent

    @property
    def recurrent_hidden_state_size(self):
        return self.base.recurrent_hidden_state_size

    def act(self, inputs, rnn_hxs, masks, deterministic=False):
        value, feats, rnn_hxs = self.base(inputs, rnn_hxs, masks)
        dist = self.dist(feats)
        action = dist.mode() if deterministic else dist.sample()
        return value, action, dist.log_probs(action), rnn_hxs

    def get_value(self, inputs, rnn_hxs, masks):
        return self.base(inputs, rnn_hxs, masks)[0]

    def evaluate_actions(self, inputs, rnn_hxs, masks, action):
        value, feats, rnn_hxs = self.base(inputs, rnn_hxs, masks)
        dist = self.dist(feats)
        return value, dist.log_probs(action), dist.entropy().mean(), rnn_hxs


class BipedalWalkerAdversaryPolicy(DeviceAwareModule):
    """The level builder (walker_models.py:170-257): the design vector, random_z and the time step concatenated into the same
    two-tower MLP, a diagonal Gaussian over the design parameters.  The continuous form every shipped config uses; the
    reference's `editor=True` branch builds a Python list of Categoricals that its own `act` cannot sample from, so it is
    not reproduced.  Kept quirks: `act` squashes the sampled action with tanh and returns the Gaussian log-prob OF THE
    SQUASHED action (:236-238); the random builder draws from numpy's global stream (:222)."""

    def __init__(self, observation_space, action_space, editor=False, random=False, base_kwargs=None):
        super().__init__()
        if editor:
            raise NotImplementedError("BipedalWalkerAdversaryPolicy(editor=True): not functional in the reference either")
        self.random = random
        self.design_dim = observation_space["image"].shape[0]
        self.random_z_dim = observation_space["random_z"].shape[0]
        self.base = MLPBase(self.design_dim + self.random_z_dim + 1, **(base_kwargs or {}))
        self.editor = False
        self.action_dim = action_space.shape[0]
        self.dist = DiagGaussian(self.base.output_size, self.action_dim)

    @property
    def is_recurrent(self):
        return self.base.is_recurrent

    @property
    def recurrent_hidden_state_size(self):
        return self.base.recurrent_hidden_state_size

    def preprocess(self, inputs):
        return torch.cat([inputs["image"], inputs["random_z"], inputs["time_step"]], dim=1)

    def act(self, inputs, rnn_hxs, masks, deterministic=False):
        x = self.preprocess(inputs)
        if self.random:
            n = x.shape[0]
            action = torch.tensor(np.random.uniform(-1, 1, n), device=x.device).reshape(-1, 1)
            return torch.zeros(n, 1, device=x.device), action, torch.ones(n, self.action_dim, device=x.device), rnn_hxs
        value, feats, rnn_hxs = self.base(x, rnn_hxs, masks)
        dist = self.dist(feats)
        action = torch.tanh(dist.mode() if deterministic else dist.sample())
        return value, action, dist.log_probs(action), rnn_hxs

    def get_value(self, inputs, rnn_hxs, masks):
        return self.base(self.preprocess(inputs), rnn_hxs, masks)[0]

    def evaluate_actions(self, inputs, rnn_hxs, masks, action):
        value, feats, rnn_hxs = self.base(self.preprocess(inputs), rnn_hxs, masks)
        dist = self.dist(feats)
        return value, dist.log_probs(action), dist.entropy().mean(), rnn_hxs
