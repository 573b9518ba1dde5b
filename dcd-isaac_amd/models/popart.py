"""PopArt value head (reference: models/popart.py:20-103; BASELINE configs 4-5 — CarRacing / BipedalWalker — train with
--use_popart).

A Linear(input_shape -> output_shape) whose targets are normalised by running moments: `update(returns)` moves the
exponential moments (mean, mean of squares, debiasing term), `normalize` / `denormalize` use the DEBIASED moments with the
variance clamped at 1e-2.  State-dict names (weight, bias, stddev, mean, mean_sq, debiasing_term — all nn.Parameters, the
last four without gradient) are the reference's, so its checkpoints load.

Reference quirk kept on purpose (pinned by tests/golden/popart.npz): `update` keeps ALIASES of the old mean / stddev
(`old_mean, old_stddev = self.mean, self.stddev`, popart.py:67) and then moves the moments IN PLACE, so the "preserve the
outputs" rescaling that follows sees the NEW values on both sides: weight <- (weight * stddev) / stddev and
bias <- (stddev * bias + mean - mean) / stddev — an identity up to rounding, evaluated here in the same order.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .common import DeviceAwareModule


class PopArt(DeviceAwareModule):
    def __init__(self, input_shape, output_shape, norm_axes=1, beta=0.99999, epsilon=1e-5):
        super().__init__()
        self.beta, self.epsilon, self.norm_axes = beta, epsilon, norm_axes
        self.input_shape, self.output_shape = input_shape, output_shape
        self.weight = nn.Parameter(torch.empty(output_shape, input_shape))
        self.bias = nn.Parameter(torch.empty(output_shape))
        self.stddev = nn.Parameter(torch.ones(output_shape), requires_grad=False)
        self.mean = nn.Parameter(torch.zeros(output_shape), requires_grad=False)
        self.mean_sq = nn.Parameter(torch.zeros(output_shape), requires_grad=False)
        self.debiasing_term = nn.Parameter(torch.tensor(0.0), requires_grad=False)
        self.reset_parameters()

    def reset_parameters(self):                                   # nn.Linear's default init (popart.py:44-52)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(self.weight.size(1))
        nn.init.uniform_(self.bias, -bound, bound)
        with torch.no_grad():
            self.mean.zero_()
            self.mean_sq.zero_()
            self.debiasing_term.zero_()

    def _tensor(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        return x.to(self.device)

    def forward(self, x):
        return F.linear(self._tensor(x), self.weight, self.bias)

    @torch.no_grad()
    def update(self, returns, group=None):
        """`group` (not in the reference: one process there): the batch moments are averaged over its ranks — equal minibatch
        sizes on every rank, so the mean of the per-rank means is the mean over the job-wide minibatch — and every rank moves
        its (replicated) PopArt state by the same amount.  One rank: x / 1, the plain update bit for bit."""
        x = self._tensor(returns)
        axes = tuple(range(self.norm_axes))
        keep = 1.0 - self.beta
        batch_mean, batch_mean_sq = x.mean(dim=axes), (x ** 2).mean(dim=axes)
        if group is not None:
            import torch.distributed as dist
            mom = torch.stack([batch_mean, batch_mean_sq])
            dist.all_reduce(mom, group=group)
            mom /= dist.get_world_size(group)
            batch_mean, batch_mean_sq = mom[0], mom[1]
        self.mean.mul_(self.beta).add_(batch_mean * keep)
        self.mean_sq.mul_(self.beta).add_(batch_mean_sq * keep)
        self.debiasing_term.mul_(self.beta).add_(1.0 * keep)
        self.stddev.data = (self.mean_sq - self.mean ** 2).sqrt().clamp(min=1e-4)
        # the reference's output-preserving step with its aliased "old" moments (see module docstring)
        self.weight.data = self.weight * self.stddev / self.stddev
        self.bias.data = (self.stddev * self.bias + self.mean - self.mean) / self.stddev

    def debiased_mean_var(self):
        d = self.debiasing_term.clamp(min=self.epsilon)
        mean, mean_sq = self.mean / d, self.mean_sq / d
        return mean, (mean_sq - mean ** 2).clamp(min=1e-2)

    def normalize(self, x):
        mean, var = self.debiased_mean_var()
        return (self._tensor(x) - mean) / torch.sqrt(var)

    def denormalize(self, x):
        mean, var = self.debiased_mean_var()
        return self._tensor(x) * torch.sqrt(var) + mean
