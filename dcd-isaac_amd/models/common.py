"""Shared network pieces for the MultiGrid policy/value nets (PyTorch-ROCm; not a HIP-kernel path).

Parameter names, construction order and initialisers follow the reference so that (a) the same
torch seed gives bit-identical initial weights and (b) reference checkpoints load unchanged:
models/common.py:18-46 (init helpers), :109-209 (masked RNN), :211-224 (one_hot, fc stack).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _orthogonal(module, gain=1.0):
    nn.init.orthogonal_(module.weight.data, gain=gain)
    nn.init.constant_(module.bias.data, 0)
    return module


def init_(m):                 # orthogonal, gain 1 (common.py:23-24)
    return _orthogonal(m, 1.0)


def init_relu_(m):
    return _orthogonal(m, nn.init.calculate_gain("relu"))


def init_tanh_(m):            # orthogonal, gain sqrt(2) (common.py:29-30)
    return _orthogonal(m, np.sqrt(2))


def apply_init_(modules, gain=None):
    """Xavier-uniform conv weights, unit norms (common.py:32-46)."""
    for m in modules:
        if isinstance(m, nn.Conv2d):
            nn.init.xavier_uniform_(m.weight, gain=gain) if gain else nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
            nn.init.constant_(m.weight, 1)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


class DeviceAwareModule(nn.Module):
    @property
    def device(self):
        return next(self.parameters()).device


class Conv2d_tf(nn.Conv2d):
    """Conv2d with TF-style 'valid' / 'same' padding (common.py:63-107).  Only 'valid' is used by
    the MultiGrid nets (multigrid_models.py:46)."""

    def __init__(self, *args, **kwargs):
        self._tf_padding = kwargs.pop("padding", "same")
        super().__init__(*args, **kwargs)

    def forward(self, x):
        if self._tf_padding == "valid":
            return F.conv2d(x, self.weight, self.bias, self.stride, 0, self.dilation, self.groups)
        pads = []
        for dim in (0, 1):
            size, k = x.size(dim + 2), self.weight.size(dim + 2)
            eff = (k - 1) * self.dilation[dim] + 1
            out = (size + self.stride[dim] - 1) // self.stride[dim]
            pads.append(max(0, (out - 1) * self.stride[dim] + eff - size))
        if pads[0] % 2 or pads[1] % 2:
            x = F.pad(x, [0, pads[1] % 2, 0, pads[0] % 2])
        return F.conv2d(x, self.weight, self.bias, self.stride, (pads[0] // 2, pads[1] // 2), self.dilation,
                        self.groups)


class RNN(nn.Module):
    """Masked LSTM/GRU (common.py:109-209): a single step multiplies the state by `masks` before the
    cell; a flattened (T*N) sequence is cut at every step where ANY env has mask 0 and the state is
    re-masked at each cut."""

    def __init__(self, input_size, hidden_size=128, arch="lstm"):
        super().__init__()
        self.arch, self.is_lstm, self._hidden_size = arch, arch == "lstm", hidden_size
        if arch == "gru":
            self.rnn = nn.GRU(input_size, hidden_size)
        elif arch == "lstm":
            self.rnn = nn.LSTM(input_size, hidden_size)
        else:
            raise ValueError(f"Unsupported RNN architecture {arch}.")
        for name, param in self.rnn.named_parameters():
            if "bias" in name:
                nn.init.constant_(param, 0)
            elif "weight" in name:
                nn.init.orthogonal_(param)

    @property
    def recurrent_hidden_state_size(self):
        return self._hidden_size

    @property
    def output_size(self):
        return self._hidden_size

    def forward(self, x, hxs, masks):
        n_state = hxs[0].size(0) if self.is_lstm else hxs.size(0)
        if x.size(0) == n_state:                                   # one step
            state = tuple((h * masks).unsqueeze(0) for h in hxs) if self.is_lstm else (hxs * masks).unsqueeze(0)
            y, state = self.rnn(x.unsqueeze(0), state)
            state = tuple(h.squeeze(0) for h in state) if self.is_lstm else state.squeeze(0)
            return y.squeeze(0), state
        N, T = n_state, x.size(0) // n_state                        # flattened sequence
        x = x.view(T, N, x.size(1))
        masks = masks.view(T, N)
        cuts = (masks[1:] == 0.0).any(dim=-1).nonzero().flatten().cpu().tolist()
        bounds = [0] + [c + 1 for c in cuts] + [T]
        state = tuple(h.unsqueeze(0) for h in hxs) if self.is_lstm else hxs.unsqueeze(0)
        outs = []
        for s, e in zip(bounds[:-1], bounds[1:]):
            m = masks[s].view(1, -1, 1)
            state = tuple(h * m for h in state) if self.is_lstm else state * m
            y, state = self.rnn(x[s:e], state)
            outs.append(y)
        y = torch.cat(outs, dim=0).view(T * N, -1)
        state = tuple(h.squeeze(0) for h in state) if self.is_lstm else state.squeeze(0)
        return y, state


def one_hot(dim, inputs, device="cpu"):
    return F.one_hot(inputs.long(), dim).squeeze(1).float()


def make_fc_layers_with_hidden_sizes(sizes, input_size):
    layers = []
    for i in range(len(sizes) - 1):
        layers += [init_tanh_(nn.Linear(input_size if i == 0 else sizes[0], sizes[i + 1])), nn.Tanh()]
    return nn.Sequential(*layers)
