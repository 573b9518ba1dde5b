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


class CarRacingBezierAdversaryEnvNetwork(DeviceAwareModule):
    """The CarRacing level builder (reference: models/car_racing_models.py:168-530; BASELINE config 5's teacher): it places the
    Bezier track's control points one per step on a 10 x 10 sketch grid.

    Trunk (:215-238): sketch [B,1,10,10] -> two 2 x 2 VALID convolutions (8, 16 filters; no non-linearity between them) ->
    flatten -> ReLU (1024) | Linear(one-hot time step) (scalar_fc) | random_z | (use_goal) Linear(one-hot goal bin).
    Heads: value (Linear or PopArt); policy either a Beta on [0,1]^action_dim (alpha, beta = 1 + softplus, :259-266) or, with
    `use_categorical`, a masked categorical over {skip} + the 100 sketch cells (occupied cells masked, the skip action masked
    unless `use_skip` allows it, :405-419); with `use_goal` the step `time_step == total_time_steps - 1` draws a goal bin from
    `goal_head` instead, and actions carry a leading goal-step flag (:436-441).

    Construction order, initialisers and parameter names are the reference's — same torch seed, same weights; its
    checkpoints load.  That includes building the value head TWICE (:241 and :244-249: the first Linear is thrown away but
    has consumed the generator).  Nothing drives this network here: the Box2D track environment has no oracle in this
    repository's scope; the module is pinned by tests/golden/f4_adv.npz (`cr_*`)."""

    def __init__(self, observation_space, action_space, scalar_fc=8, use_categorical=False, use_skip=False, choose_start_pos=False,
                 use_popart=False, set_start_pos=False, use_goal=False, num_goal_bins=1):
        super().__init__()
        from .common import Conv2d_tf, apply_init_
        self.sketch_dim = observation_space["control_points"].shape
        self.random_z_dim = observation_space["random_z"].shape[0]
        self.total_time_steps = observation_space["time_step"].high[0]
        self.time_step_dim = self.total_time_steps + 1                       # + the terminal time step
        if use_goal:
            self.goal_bin_dim = observation_space["goal_bin"].high[0]
        self.scalar_fc, self.set_start_pos, self.use_categorical, self.use_skip = scalar_fc, set_start_pos, use_categorical, use_skip
        self.use_goal, self.num_goal_bins = use_goal, num_goal_bins
        self.random, self.action_space = False, action_space
        self.n_control_points = self.time_step_dim - int(bool(use_goal)) - int(bool(set_start_pos))
        if use_categorical:
            self.action_dim = int(np.prod(self.sketch_dim)) + 1              # + the skip action
        else:
            self.action_dim = action_space.shape[0] - int(bool(use_goal))    # no Beta for the goal-flag prefix
        self.sketch_embedding = nn.Sequential(Conv2d_tf(1, 8, kernel_size=2, stride=1, padding="valid"),
                                              Conv2d_tf(8, 16, kernel_size=2, stride=1, padding="valid"), nn.Flatten(), nn.ReLU())
        self.sketch_embed_dim = 1024
        self.ts_embedding = nn.Linear(self.time_step_dim, scalar_fc)
        width = self.sketch_embed_dim + scalar_fc + self.random_z_dim
        if use_goal:
            self.goal_bin_embedding = nn.Linear(num_goal_bins + 1, scalar_fc)
            width += scalar_fc
        self.base_output_size = width
        init_(nn.Linear(width, 1))                                           # the discarded first value head (:241) — generator state
        self.critic = init_(PopArt(width, 1)) if use_popart else init_(nn.Linear(width, 1))
        self.popart = self.critic if use_popart else None
        if use_categorical:
            self.actor = nn.Sequential(init_(nn.Linear(width, 256)), nn.ReLU(), init_(nn.Linear(256, self.action_dim)))
        else:
            self.fc_alpha = nn.Sequential(init_relu_(nn.Linear(width, self.action_dim)), nn.Softplus())
            self.fc_beta = nn.Sequential(init_relu_(nn.Linear(width, self.action_dim)), nn.Softplus())
        if use_goal:
            self.goal_head = nn.Sequential(init_(nn.Linear(width, 256)), nn.ReLU(), init_(nn.Linear(256, num_goal_bins)))
        apply_init_(self.modules())
        self.train()

    is_recurrent = False
    recurrent_hidden_state_size = 1

    def forward(self, inputs, rnn_hxs, masks):
        raise NotImplementedError

    def _forward_base(self, inputs, rnn_hxs, masks):
        from .common import one_hot
        parts = [self.sketch_embedding(inputs["control_points"]),
                 self.ts_embedding(one_hot(self.time_step_dim, inputs["time_step"]).to(self.device)), inputs["random_z"]]
        if self.use_goal:
            parts.append(self.goal_bin_embedding(one_hot(self.goal_bin_dim, inputs["goal_bin"]).to(self.device)))
        return torch.cat(parts, dim=-1)

    def process_action(self, action):
        """Stored action -> what the env's step_adversary takes (:304-326): a goal bin on a goal step, else (x, y, skip)."""
        import torch.nn.functional as F
        if self.use_goal:
            if action[0][0]:
                return action[:, 1]
            action = action[:, 1:]
        if self.use_categorical:
            x = ((action - 1.0) % self.sketch_dim[-1]) / self.sketch_dim[-1]
            y = ((action - 1.0) // self.sketch_dim[-2]) / self.sketch_dim[-2]
            return torch.cat((x, y, (action == 0).float()), dim=-1)
        skip = F.gumbel_softmax(torch.log(action[:, -1] + 1e-5), tau=1, hard=True).unsqueeze(-1)
        return torch.cat((action[:, :-1], skip), dim=-1)

    def _sketch_to_mask(self, sketch):                                      # column 0 = skip, then the occupied cells
        return torch.cat((torch.zeros(sketch.shape[0], 1, dtype=torch.bool, device=self.device), sketch.flatten(1).bool()), dim=-1)

    def _is_goal_step(self, t):
        return t == self.total_time_steps - 1 if self.use_goal else False

    def _is_start_pos_step(self, t):
        return t == self.n_control_points if self.set_start_pos else False

    def _with_goal_flag(self, action, is_goal_step):
        if not self.use_goal:
            return action
        first = action[:, 0].unsqueeze(-1)
        return torch.cat((torch.ones_like(first) if is_goal_step else torch.zeros_like(first), action), dim=-1)

    def act_random(self, inputs, rnn_hxs):
        """:347-392 (categorical form; the reference's continuous branch reads an attribute it never sets)."""
        is_goal_step = self._is_goal_step(inputs["time_step"][0])
        B = inputs["time_step"].shape[0]
        values = torch.zeros((B, 1), device=self.device)
        action_log_dist = torch.ones((B, 1), device=self.device)
        shape = self.action_space.shape
        if self.use_goal:
            shape = (shape[0] - 1,)
        if is_goal_step:
            action = torch.zeros((B, 1), dtype=torch.int64, device=self.device)
            for b in range(B):
                action[b] = np.random.randint(self.num_goal_bins)
        else:
            if not self.use_categorical:
                raise NotImplementedError("act_random of the continuous form reads `self.action_shape`, which the reference never defines")
            action = torch.zeros(B, *shape, dtype=torch.int64, device=self.device)
            high = self.action_space.high[1] if self.use_goal else self.action_space.high[0]
            for b in range(B):
                action[b] = np.random.randint(1, high)                       # never the skip action 0
        return values, self._with_goal_flag(action, is_goal_step), action_log_dist, rnn_hxs

    def _masked_logits(self, embedded, mask, time_step=None, start_pos_rows=None):
        """Categorical logits with occupied cells (and, unless allowed, the skip action) at the dtype's minimum."""
        logits = self.actor(embedded)
        if self.use_skip:
            if start_pos_rows is not None:                                   # evaluate_actions (:489-492)
                mask[mask.sum(-1) < 3, 0] = True
                logits[start_pos_rows] = torch.finfo(logits.dtype).min
            elif self._is_start_pos_step(time_step):                         # act (:410-413): the start position cannot be skipped
                mask[:, 0] = True
            else:
                mask[mask.sum(-1) < 3, 0] = True
        else:
            mask[:, 0] = True
        logits[mask] = torch.finfo(logits.dtype).min
        return logits

    def act(self, inputs, rnn_hxs, masks, deterministic=False):
        if self.random:
            return self.act_random(inputs, rnn_hxs)
        from .distributions import FixedCategorical
        e = self._forward_base(inputs, rnn_hxs, masks)
        value = self.critic(e)
        t0 = inputs["time_step"][0]
        is_goal_step = self._is_goal_step(t0)
        if is_goal_step:
            dist = FixedCategorical(logits=self.goal_head(e))
            action = dist.sample()
            logp = dist.log_probs(action)
        elif self.use_categorical:
            dist = FixedCategorical(logits=self._masked_logits(e, self._sketch_to_mask(inputs["control_points"]), time_step=t0))
            action = dist.sample()
            logp = dist.log_probs(action)
        else:
            dist = Beta(1 + self.fc_alpha(e), 1 + self.fc_beta(e))
            action = dist.sample()
            logp = dist.log_prob(action).sum(dim=1).unsqueeze(1)
        return value, self._with_goal_flag(action, is_goal_step), logp, rnn_hxs      # log-dist slot = log-probs (:433-434)

    def get_value(self, inputs, rnn_hxs, masks):
        return self.critic(self._forward_base(inputs, rnn_hxs, masks))

    def evaluate_actions(self, inputs, rnn_hxs, masks, action):
        from .distributions import FixedCategorical
        B = len(inputs["time_step"])
        e = self._forward_base(inputs, rnn_hxs, masks)
        value = self.critic(e)
        time_steps = inputs["time_step"]
        mask = self._sketch_to_mask(inputs["control_points"])
        if self.use_goal:
            action = action[:, 1:]
            goal_rows = self._is_goal_step(time_steps).flatten()
            any_goal, any_other = bool(goal_rows.any()), bool((~goal_rows).any())
        else:
            goal_rows, any_goal, any_other = False, False, True
        body = e
        if any_goal:
            goal_dist = FixedCategorical(logits=self.goal_head(e[goal_rows]))
            goal_logp = goal_dist.log_probs(action[goal_rows][0])           # (:475: the first goal row's action, as written)
            body, mask, action = e[~goal_rows], mask[~goal_rows], action[~goal_rows]
        if any_other:
            if self.use_categorical:
                rows = self._is_start_pos_step(time_steps[~goal_rows]) if self.use_skip else None
                dist = FixedCategorical(logits=self._masked_logits(body, mask, start_pos_rows=rows if self.use_skip else None))
                logp = dist.log_probs(action)
            else:
                dist = Beta(1 + self.fc_alpha(body), 1 + self.fc_beta(body))
                logp = dist.log_prob(action).sum(dim=1).unsqueeze(1)
        if not self.use_goal:
            return value, logp, dist.entropy().mean(), rnn_hxs
        both = torch.zeros((B, 1), dtype=torch.float, device=self.device)
        entropy = 0
        if any_goal:
            both[goal_rows] = goal_logp
            entropy = entropy + goal_dist.entropy().sum()
        if any_other:
            both[~goal_rows] = logp
            entropy = entropy + dist.entropy().sum()
        return value, both, entropy / B, rnn_hxs
