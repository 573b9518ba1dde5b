"""CarRacing student network (reference: models/car_racing_models.py:18-165 CarRacingNetwork; BASELINE config 5, SURVEY
§8f-4): six stride-2 / valid convolutions down to a 256-vector (:51-64 for the 96 x 96 input, :37-50 for the cropped
84 x 84 one), a ReLU hidden layer per head, a Beta policy on [0,1]^3 (alpha, beta = 1 + softplus) rescaled to the action
box by `process_action`, a linear or PopArt value head.  PyTorch-ROCm; parameter names, construction order and
initialisers are the reference's (same seed -> same weights, its checkpoints load).  The Box2D car dynamics have no
oracle in this environment (parity unpinned)."""
import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Beta

from .common import DeviceAwareModule, init_, init_relu_
from .popart import PopArt

# (out_channels, kernel, stride) per layer
_CONVS = {False: [(8, 4, 2), (16, 3, 2), (32, 3, 2), (64, 3, 2), (128, 3, 1), (256, 3, 1)],      # 96 x 96
          True: [(8, 2, 2), (16, 2, 2), (32, 2, 2), (64, 2, 2), (128, 3, 1), (256, 3, 1)]}       # cropped: 84 x 84


class CarRacingNetwork(DeviceAwareModule):
    def __init__(self, obs_shape, action_space, hidden_size=100, crop=False, use_popart=False):
        super().__init__()
        self.action_low = np.array(action_space.low, dtype=np.float32)
        self.action_high = np.array(action_space.high, dtype=np.float32)
        self.action_dim = action_space.shape[0]
        self.crop = crop
        layers, c = [], obs_shape[-3]
        for out, k, s in _CONVS[bool(crop)]:
            layers += [nn.Conv2d(c, out, kernel_size=k, stride=s), nn.ReLU()]
            c = out
        self.image_embed = nn.Sequential(*layers)
        self.image_embedding_size = 256
        self.actor_fc = nn.Sequential(init_relu_(nn.Linear(256, hidden_size)), nn.ReLU())
        self.actor_alpha = nn.Sequential(init_relu_(nn.Linear(hidden_size, self.action_dim)), nn.Softplus())
        self.actor_beta = nn.Sequential(init_relu_(nn.Linear(hidden_size, self.action_dim)), nn.Softplus())
        value_out = init_(PopArt(hidden_size, 1)) if use_popart else init_(nn.Linear(hidden_size, 1))
        self.popart = value_out if use_popart else None
        self.critic = nn.Sequential(init_relu_(nn.Linear(256, hidden_size)), nn.ReLU(), value_out)
        for m in self.modules():                                   # :113-116: convs xavier (relu gain), bias 0.1
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight, gain=nn.init.calculate_gain("relu"))
                nn.init.constant_(m.bias, 0.1)
        self.train()

    is_recurrent = False
    recurrent_hidden_state_size = 1

    def process_action(self, action):
        return action * (self.action_high - self.action_low) + self.action_low

    def _heads(self, inputs):
        e = self.image_embed(inputs).squeeze()
        h = self.actor_fc(e)
        return e, Beta(1 + self.actor_alpha(h), 1 + self.actor_beta(h), validate_args=False)

    def act(self, inputs, rnn_hxs, masks, deterministic=False):
        e, dist = self._heads(inputs)
        action = dist.sample()
        log_dist = dist.log_prob(action).sum(dim=-1).unsqueeze(-1)
        if inputs.shape[0] == 1:
            action = action.unsqueeze(0)
        return self.critic(e), action, log_dist, rnn_hxs

    def get_value(self, inputs, rnn_hxs, masks):
        return self.critic(self.image_embed(inputs).squeeze())

    def evaluate_actions(self, inputs, rnn_hxs, masks, action):
        e, dist = self._heads(inputs)
        return self.critic(e), dist.log_prob(action).sum(dim=-1).unsqueeze(-1), dist.entropy().mean(), rnn_hxs
