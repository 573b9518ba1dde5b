"""Categorical policy head (reference: models/distributions.py:18-56)."""
import torch
import torch.nn as nn


class FixedCategorical(torch.distributions.Categorical):
    def sample(self):
        p = self.probs
        if p.is_cuda and torch.cuda.is_current_stream_capturing():
            # Inside a HIP-graph capture the draw must not go through torch.multinomial: its validity check is built from
            # the full reductions p.max() / p.min(), and PyTorch's multi-block full reduction (per-block partials combined
            # by the last block behind a semaphore) returns results assembled from STALE partials when it is replayed
            # from a HIP graph on this stack (PyTorch 2.10 + ROCm 7.2, gfx950; tools/graph_node_order_probe.py: ~0.2 % of
            # the replayed reductions, none eagerly).  A "minimum" made of the previous user's partials — the logits, which
            # are negative — fails multinomial's `_assert_async`, which traps the queue and aborts the process: the
            # round-1 abort.  The draw below is what multinomial itself computes for one sample per row
            # (argmax of p / Exp(1), row-wise: single-pass reductions only), minus the assertions.
            q = torch.empty_like(p).exponential_(1)
            return (p / q).argmax(dim=-1, keepdim=True)
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
