"""Categorical policy head (reference: models/distributions.py:18-56)."""
import torch
import torch.nn as nn


class FixedCategorical(torch.distributions.Categorical):
    def sample(self):
        return super().sample().unsqueeze(-1)

    def log_probs(self, actions):
        return super().log_prob(actions.squeeze(-1)).view(actions.size(0), -1).sum(-1).unsqueeze(-1)

    def mode(self):
        return self.probs.argmax(dim=-1, keepdim=True)


class Categorical(nn.Module):
    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        from .common import Linear
        self.linear = Linear(num_inputs, num_outputs)
        nn.init.orthogonal_(self.linear.weight.data, gain=0.01)
        nn.init.constant_(self.linear.bias.data, 0)

    def forward(self, x):
        # validate_args=False: the constraint check synchronises the stream (and is illegal inside HIP-graph capture)
        return FixedCategorical(logits=self.linear(x), validate_args=False)
