"""MultigridNetwork — student (5x5 view) and teacher (15x15 grid + time step + z) actor-critic
(reference: models/multigrid_models.py:15-186; sizes from util/make_agent.py:15-58).

Stays in PyTorch-ROCm by design (BASELINE north_star): conv 3->F (3x3, valid) + ReLU, one-hot
scalar embedding, optional random_z, masked LSTM(256), tanh MLP heads.  Module names/ordering match
the reference so seeds and checkpoints are interchangeable.
"""
import torch
import torch.nn as nn

from .common import (Conv2d_tf, DeviceAwareModule, RNN, apply_init_, init_, make_fc_layers_with_hidden_sizes,
                     one_hot)
from .distributions import Categorical


class MultigridNetwork(DeviceAwareModule):
    def __init__(self, observation_space, action_space, actor_fc_layers=(32, 32), value_fc_layers=(32, 32),
                 conv_filters=16, conv_kernel_size=3, scalar_fc=5, scalar_dim=4, random_z_dim=0, xy_dim=0,
                 recurrent_arch="lstm", recurrent_hidden_size=256, random=False):
        super().__init__()
        self.random = random
        self.action_space = action_space
        shape = observation_space["image"].shape
        c, m, n = shape[-3], shape[-2], shape[-1]
        self.image_conv = nn.Sequential(
            Conv2d_tf(c, conv_filters, kernel_size=conv_kernel_size, stride=1, padding="valid"),
            nn.Flatten(), nn.ReLU())
        self.image_embedding_size = (n - conv_kernel_size + 1) * (m - conv_kernel_size + 1) * conv_filters
        self.preprocessed_input_size = self.image_embedding_size
        self.xy_embed, self.xy_dim = None, xy_dim
        if xy_dim:
            self.preprocessed_input_size += 2 * xy_dim
        self.scalar_embed, self.scalar_dim = None, scalar_dim
        if scalar_dim:
            self.scalar_embed = nn.Linear(scalar_dim, scalar_fc)
            self.preprocessed_input_size += scalar_fc
        self.preprocessed_input_size += random_z_dim
        self.base_output_size = self.preprocessed_input_size
        self.rnn = None
        if recurrent_arch:
            self.rnn = RNN(self.preprocessed_input_size, recurrent_hidden_size, arch=recurrent_arch)
            self.base_output_size = recurrent_hidden_size
        self.actor = nn.Sequential(make_fc_layers_with_hidden_sizes(actor_fc_layers, self.base_output_size),
                                   Categorical(actor_fc_layers[-1], action_space.n))
        self.critic = nn.Sequential(make_fc_layers_with_hidden_sizes(value_fc_layers, self.base_output_size),
                                    init_(nn.Linear(value_fc_layers[-1], 1)))
        apply_init_(self.modules())
        self.train()

    @property
    def is_recurrent(self):
        return self.rnn is not None

    @property
    def recurrent_hidden_state_size(self):
        return self.rnn.recurrent_hidden_state_size if self.rnn is not None else 0

    def forward(self, inputs, rnn_hxs, masks):
        raise NotImplementedError

    def _forward_base(self, inputs, rnn_hxs, masks):
        scalar = inputs.get("direction")
        if scalar is None:
            scalar = inputs.get("time_step")
        parts = [self.image_conv(inputs.get("image"))]
        if self.scalar_embed is not None:
            parts.append(self.scalar_embed(one_hot(self.scalar_dim, scalar).to(self.device)))
        z = inputs.get("random_z")
        if z is not None:
            parts.append(z)
        feats = torch.cat(parts, dim=-1)
        if self.rnn is not None:
            feats, rnn_hxs = self.rnn(feats, rnn_hxs, masks)
        return feats, rnn_hxs

    def act(self, inputs, rnn_hxs, masks, deterministic=False):
        if self.random:                                              # random teacher/student (:149-157)
            B = inputs["image"].shape[0]
            action = torch.randint(0, self.action_space.n, (B, 1), dtype=torch.int64, device=self.device)
            values = torch.zeros((B, 1), device=self.device)
            action_log_dist = torch.ones(B, self.action_space.n, device=self.device)
            return values, action, action_log_dist, rnn_hxs
        feats, rnn_hxs = self._forward_base(inputs, rnn_hxs, masks)
        dist = self.actor(feats)
        value = self.critic(feats)
        action = dist.mode() if deterministic else dist.sample()
        return value, action, dist.logits, rnn_hxs

    def get_value(self, inputs, rnn_hxs, masks):
        feats, _ = self._forward_base(inputs, rnn_hxs, masks)
        return self.critic(feats)

    def evaluate_actions(self, inputs, rnn_hxs, masks, action):
        feats, rnn_hxs = self._forward_base(inputs, rnn_hxs, masks)
        dist = self.actor(feats)
        return self.critic(feats), dist.log_probs(action), dist.entropy().mean(), rnn_hxs
