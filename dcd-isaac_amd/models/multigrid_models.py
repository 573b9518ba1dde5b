"""MultigridNetwork — student (5x5 view) and teacher (15x15 grid + time step + z) actor-critic
(reference: models/multigrid_models.py:15-186; sizes from util/make_agent.py:15-58).

Stays in PyTorch-ROCm by design (BASELINE north_star): conv 3->F (3x3, valid) + ReLU, one-hot
scalar embedding, optional random_z, masked LSTM(256), tanh MLP heads.  Module names/ordering match
the reference so seeds and checkpoints are interchangeable.
"""
import torch
import torch.nn as nn

from .common import (Conv2d_tf, DeviceAwareModule, Linear, RNN, apply_init_, init_, linear,
                     make_fc_layers_with_hidden_sizes, one_hot)
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
            self.scalar_embed = Linear(scalar_dim, scalar_fc)
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
                                    init_(Linear(value_fc_layers[-1], 1)))
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

    # ---- incremental input projection for grid-building sequences (the teacher's PPO batch) ----------------
    # A constructive teacher places ONE object per step, so consecutive observations differ in one or two cells and
    # the conv features F_t (F x P values, 21 632 for the 15 x 15 teacher) differ from F_{t-1} in at most 9 of the P
    # output positions.  The LSTM's input projection is linear in the features, hence
    #     F_t W^T = F_{t-1} W^T + (F_t - F_{t-1}) W^T ,
    # and the right-hand term needs only the changed positions: per position p a [rows_p, F] x [F, 4H] product.  The
    # batch's projection becomes one dense GEMM for t = 0 plus ~5 % of the dense FLOPs, accumulated over time with a
    # cumsum; autograd differentiates the same composition (the backward of the cumsum is the reverse cumsum of the
    # incoming gradient, which turns the two backward GEMMs sparse in the same way).  Exact algebra — results equal the
    # dense path up to fp32 summation order (tests/test_gpu_policy.py); falls back to the dense path when more than
    # `INCREMENTAL_MAX_DENSITY` of the (t, n, p) triples changed.
    INCREMENTAL_MAX_PADDING = 2.0      # batched form only while the padded [P, cmax] layout is at most this x the triples
    INCREMENTAL_MAX_DENSITY = 0.3

    def _incremental_eligible(self, image, n_state):
        import os
        return (image is not None and image.is_cuda and self.rnn is not None and self.rnn.is_lstm
                and image.dim() == 4 and image.size(0) >= 4 * n_state and image.size(0) % n_state == 0
                and self.image_embedding_size >= 4096 and self.image_conv[0].kernel_size == (3, 3)
                and os.environ.get("DCD_INCREMENTAL_PROJECTION", "1") != "0")

    def _incremental_projection(self, image, N):
        """xp_conv [T*N, 4H] = relu(conv(image)).flatten(1) @ W_ih[:, :F*P]^T for a time-major [T*N, 3, S, S] batch of N
        sequences, or None when the sequences are not sparse enough."""
        import torch.nn.functional as Fn
        conv = self.image_conv[0]
        lstm = self.rnn.rnn
        TN, C, S, _ = image.shape
        T, O, F = TN // N, S - 2, conv.out_channels
        P, G = O * O, lstm.weight_ih_l0.size(0)
        img = image.view(T, N, C, S, S)
        with torch.no_grad():
            changed = (img[1:] != img[:-1]).any(dim=2)                                      # [T-1,N,S,S]
            dirty = Fn.max_pool2d(changed.float().view(-1, 1, S, S), 3, 1).view(T - 1, N, O, O) > 0
            tt, nn_, oy, ox = dirty.nonzero(as_tuple=True)
            M = tt.numel()
            if M > self.INCREMENTAL_MAX_DENSITY * (T - 1) * N * P:
                return None
            p = oy * O + ox
            order = torch.argsort(p, stable=True)                                           # group the triples by position
            counts_dev = torch.bincount(p, minlength=P)
            # nonzero() lists the triples of one (t, n) contiguously: the k-th triple of a row is at distance k from the
            # first one of its run (left insertion point of the row key, which is non-decreasing)
            key = (tt + 1) * N + nn_
            rank = torch.arange(M, device=image.device) - torch.searchsorted(key, key)
            same0 = (img[0] == img[0, :1]).all()                                            # every sequence starts from one frame
            host = torch.cat([counts_dev, rank.max().view(1) if M else counts_dev.new_zeros(1), same0.view(1).long()]).cpu().tolist()
            counts, kmax, same0 = host[:P], host[P] + 1, bool(host[P + 1])                  # (one host sync)
            tt, nn_, oy, ox = tt[order], nn_[order], oy[order], ox[order]
            row_t, row_prev = (tt + 1) * N + nn_, tt * N + nn_
            cmax = max(counts) if counts else 0
            batched = M > 0 and cmax * P <= self.INCREMENTAL_MAX_PADDING * M
            if batched:                                      # slot of triple i in the [P, cmax] padded layout
                starts = torch.cumsum(counts_dev, 0) - counts_dev
                p_sorted = p[order]
                slot = p_sorted * cmax + (torch.arange(M, device=image.device) - starts[p_sorted])
                row_of = torch.zeros((P * cmax,), dtype=torch.long, device=image.device)   # padding -> row 0 (unused grad)
                row_of[slot] = row_t
                src = torch.full((TN, kmax), -1, dtype=torch.int32, device=image.device)   # src[r, k]: slot of row r's k-th triple
                slot_u = torch.empty_like(slot)
                slot_u[order] = slot
                src[key, rank] = slot_u.to(torch.int32)
            k = torch.arange(3, device=image.device)
            offs = (torch.arange(C, device=image.device).view(C, 1, 1) * S * S + k.view(1, 3, 1) * S + k.view(1, 1, 3)).reshape(-1)
            col = (oy * S + ox).unsqueeze(1) + offs.unsqueeze(0)                            # [M, 27]: the 3x3xC window of each triple
        flat = image.reshape(TN, C * S * S)
        wc, w_ih = conv.weight.view(F, -1), lstm.weight_ih_l0
        wf = w_ih[:, :F * P]
        xp = torch.zeros((TN, G), device=image.device, dtype=image.dtype)
        if same0:                                       # the teacher's first frame is the same empty grid in every env:
            xp[:N] = (self.image_conv(img[0, :1]) @ wf.t()).expand(N, G)                     # one row, not a [N, 21632] GEMM
        else:
            xp[:N] = self.image_conv(img[0]) @ wf.t()                                        # t = 0: dense
        if M:
            # (`linear`: on >= 65 536 rows the weight gradient is a split-K contraction, models/common.py::matmul_tn)
            act = lambda rows: torch.relu(linear(flat[rows.unsqueeze(1), col], wc, conv.bias))
            delta = act(row_t) - act(row_prev)                                               # [M, F] feature change per triple
        if M and batched:
            # One batched GEMM over the positions instead of P small ones (545 launches per epoch, host-bound and at
            # ~1/3 of the large-GEMM rate): the triples of a position are padded to the longest group (teacher cells
            # are near-uniform: ~10 % padding), [P, cmax, F] x [P, F, 4H]; each row then sums its <= 18 product rows
            # in a fixed order (dcd_gather_sum_rows: no atomics; one atomic index_add_ over the 4 GB product took 8 ms).
            from .common import GatherSumRows
            dpad = delta.new_zeros((P * cmax, F)).index_copy(0, slot, delta)
            wp = wf.view(G, F, P).permute(2, 1, 0).contiguous()                              # [P, F, 4H]
            xp = GatherSumRows.apply(xp, torch.bmm(dpad.view(P, cmax, F), wp).view(P * cmax, G), src, row_of)
        elif M:
            # unbind / split (ONE backward node each, which stacks / concatenates the pieces' grads): indexing wperm[q]
            # and delta[a:b] inside the loop would make every iteration's backward materialise a full-size zero tensor
            wparts = wf.view(G, F, P).permute(2, 1, 0).contiguous().unbind(0)               # P x [F, 4H]
            dparts, rparts = delta.split(counts), row_t.split(counts)
            for q, c in enumerate(counts):                                                   # rows of one position are distinct
                if c:                                                                        # (t, n): the scatter-add is deterministic
                    xp.index_add_(0, rparts[q], dparts[q] @ wparts[q])
        return xp.view(T, N, G).cumsum(0).view(TN, G)

    def _forward_base(self, inputs, rnn_hxs, masks):
        scalar = inputs.get("direction")
        if scalar is None:
            scalar = inputs.get("time_step")
        image = inputs.get("image")
        n_state = rnn_hxs[0].size(0) if isinstance(rnn_hxs, (tuple, list)) else (rnn_hxs.size(0) if rnn_hxs is not None else 0)
        if self.rnn is not None and n_state and self._incremental_eligible(image, n_state):
            xp = self._incremental_projection(image, n_state)
            if xp is not None:
                lstm = self.rnn.rnn
                rest = []
                if self.scalar_embed is not None:
                    rest.append(self.scalar_embed(one_hot(self.scalar_dim, scalar).to(self.device)))
                z = inputs.get("random_z")
                if z is not None:
                    rest.append(z)
                xp = xp + (lstm.bias_ih_l0 + lstm.bias_hh_l0)
                if rest:
                    xp = xp + linear(torch.cat(rest, dim=-1), lstm.weight_ih_l0[:, self.image_embedding_size:])
                return self.rnn.forward_projected(xp, rnn_hxs, masks)
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

    def _fused_ppo_eligible(self, inputs, rnn_hxs, action):
        """The PPO batch of the registered student on the device: features / heads as fused HIP kernels (models/fused_ppo.py)."""
        from . import fused_ppo
        ok = getattr(self, "_fused_ppo_supported", None)
        if ok is None:
            ok = self._fused_ppo_supported = fused_ppo.supported(self)
        if not ok or not fused_ppo.enabled() or not isinstance(rnn_hxs, (tuple, list)) or action.dtype != torch.int64:
            return False
        image = inputs["image_u8"] if "image_u8" in inputs else inputs.get("image")
        n_state = rnn_hxs[0].size(0)
        return (image is not None and image.is_cuda and image.dim() == 4 and n_state > 0 and image.size(0) % n_state == 0
                and self.image_conv[0].weight.is_cuda)

    def evaluate_actions(self, inputs, rnn_hxs, masks, action):
        if self._fused_ppo_eligible(inputs, rnn_hxs, action):
            from . import fused_ppo
            return fused_ppo.evaluate_actions(self, inputs, rnn_hxs, masks, action)
        feats, rnn_hxs = self._forward_base(inputs, rnn_hxs, masks)
        dist = self.actor(feats)
        return self.critic(feats), dist.log_probs(action), dist.entropy().mean(), rnn_hxs
