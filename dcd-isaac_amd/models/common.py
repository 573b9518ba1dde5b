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


import os as _os
MATMUL_TN_MIN_ROWS = int(_os.environ.get("DCD_MATMUL_TN_MIN_ROWS", "65536"))


def matmul_tn(a, b):
    """a^T @ b for tall operands ([R, m]^T @ [R, k], R ~ 10^6): the weight-gradient shape of every layer in the PPO
    batch.  rocBLAS runs it as one skinny GEMM with a 10^6-long reduction (1.96 ms for m = 32, k = 256; 5.6 ms for the
    conv's m = 16, k = 27 over 9.4M rows); as a batched GEMM over S slices of the rows followed by a sum over the S
    partial products (split-K) the same contraction takes 0.25 / 0.28 ms (tools/splitk_probe.py).  Same fp32 products,
    partial sums added in a different order."""
    R = a.size(0)
    if not a.is_cuda or R < MATMUL_TN_MIN_ROWS:
        return a.t() @ b
    S = 256
    while S > 1 and (R % S or R // S < 256):
        S //= 2
    if S == 1:
        # no power-of-two split divides R (the teacher's 1 265 777 changed-window rows): split the largest multiple of 256 rows
        # 256 ways and add the product of the < 256 rows left over (round 4: the plain skinny GEMM ran [128 x 1.27M] x [1.27M x 27]
        # at 4.7 TFLOP/s, 1.9 ms a call, ten calls per teacher update)
        S = 256
        main = (R // S) * S
        if main // S < 256:
            return a.t() @ b
        out = torch.bmm(a[:main].reshape(S, main // S, a.size(1)).transpose(1, 2), b[:main].reshape(S, main // S, b.size(1))).sum(0)
        return out + a[main:].t() @ b[main:] if main < R else out
    return torch.bmm(a.reshape(S, R // S, a.size(1)).transpose(1, 2), b.reshape(S, R // S, b.size(1))).sum(0)


_ONES = {}


def _ones_column(g):
    key = (g.size(0), g.device, g.dtype)
    t = _ONES.get(key)
    if t is None:
        if len(_ONES) > 8:
            _ONES.clear()
        t = _ONES[key] = torch.ones((g.size(0), 1), device=g.device, dtype=g.dtype)
    return t


class _LinearSplitK(torch.autograd.Function):
    """y = x @ w^T + b whose weight gradient goes through matmul_tn."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return torch.addmm(b, x, w.t()) if b is not None else x @ w.t()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = matmul_tn(g, x) if ctx.needs_input_grad[1] else None
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # PyTorch's column sum of a tall narrow matrix is slow ([110 592, 169]: 0.39 ms, [1 M, 7]: 0.23 ms, [1 M, 5]: 0.69 ms,
            # profiles/r04_ppo_shapes_before.log); the same sum as a split-K product with a ones vector runs at GEMM speed
            gb = matmul_tn(g, _ones_column(g)).view(-1) if g.size(1) <= 256 else g.sum(0)
        return gx, gw, gb


def linear(x, weight, bias=None):
    """F.linear for 2-D inputs; on the PPO batch (>= MATMUL_TN_MIN_ROWS = 65 536 rows on the GPU) with the split-K weight
    gradient.  (tests/test_gpu_ppo_vs_reference.py lowers the threshold so that a fixture of 1 024 rows takes this branch.)"""
    if x.is_cuda and x.dim() == 2 and x.size(0) >= MATMUL_TN_MIN_ROWS and torch.is_grad_enabled():
        return _LinearSplitK.apply(x, weight, bias)
    return F.linear(x, weight, bias)


class Linear(nn.Linear):
    """nn.Linear (same parameters, same state-dict names) routed through `linear`."""

    def forward(self, x):
        return linear(x, self.weight, self.bias)


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
            if x.is_cuda and self.groups == 1 and self.stride == (1, 1) and self.dilation == (1, 1):
                # no MIOpen on the GPU path: its per-shape kernel build stalls for seconds on every new batch size
                # (3.9 s per update for the ragged truncated-value batch) and its launch geometry overflows at T*B
                return self._valid_conv_gemm(x)
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


def _valid_conv_gemm(self, x):
    """'valid' conv as explicit im2col (kh*kw shifted window slices) + ONE GEMM.  Used for batches beyond
    MIOpen's launch geometry (the PPO batch is T*B = 1 048 576 images: 'invalid configuration argument')
    and checked against F.conv2d in tests; F.unfold is avoided (13 ms for 4096 5x5 images on ROCm)."""
    N, C, H, W = x.shape
    O, _, kh, kw = self.weight.shape
    oh, ow = H - kh + 1, W - kw + 1
    K = C * kh * kw
    outs = []
    for s in range(0, N, 262144):                                  # bound the im2col workspace
        xs = x[s:s + 262144]
        n = xs.size(0)
        cols = torch.stack([xs[:, :, i:i + oh, j:j + ow] for i in range(kh) for j in range(kw)], dim=2)  # [n,C,kh*kw,oh,ow]
        a = cols.permute(0, 3, 4, 1, 2).reshape(n * oh * ow, K)      # rows: (image, out cell); cols: (c, ki, kj)
        y = linear(a, self.weight.view(O, K), self.bias)
        outs.append(y.view(n, oh * ow, O).transpose(1, 2).reshape(n, O, oh, ow))
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


Conv2d_tf._valid_conv_gemm = _valid_conv_gemm


class GatherSumRows(torch.autograd.Function):
    """base[r] += sum_k y[src[r, k]] (src < 0: no term) in place, by `dcd_gather_sum_rows`: every y row read once, terms
    added in k order (no atomics).  `row_of` [len(y)] names the row each y row goes to (any valid row for rows that go
    nowhere: their gradient is never used) — the backward of the scatter is that gather."""

    @staticmethod
    def forward(ctx, base, y, src, row_of):
        import ctypes as C
        from dcd_hip import check, lib, ptr, stream_ptr
        R, G = base.shape
        assert base.is_contiguous() and y.is_contiguous() and src.is_contiguous() and src.dtype == torch.int32 and src.shape[0] == R
        check(lib().dcd_gather_sum_rows(ptr(y, torch.float32), ptr(src, torch.int32), ptr(base, torch.float32), C.c_int32(R),
                                        C.c_int32(src.shape[1]), C.c_int32(G), stream_ptr()))
        ctx.mark_dirty(base)
        ctx.save_for_backward(row_of)
        return base

    @staticmethod
    def backward(ctx, grad):
        (row_of,) = ctx.saved_tensors
        return grad, grad.index_select(0, row_of), None, None


def _fused_lstm_enabled():
    import os
    return os.environ.get("DCD_FUSED_LSTM", "1") != "0"


class MaskedLSTMSequence(torch.autograd.Function):
    """The masked LSTM over a whole rollout (common.py:157-207: state multiplied by masks[t] before every step)
    as ONE autograd node: per time step one rocBLAS GEMM + one HIP pointwise kernel forward
    (dcd_lstm_seq_fwd_step) and the same backward (dcd_lstm_seq_bwd_step), instead of ~14 + ~26 elementwise
    launches per step through autograd.  d W_hh and d xp come out of two GEMMs over the whole [T*N] batch.

    xp: [T,N,4H] input projection (+ both biases); h0, c0: [N,H]; masks: [T,N]; w_hh: [4H,H]."""

    @staticmethod
    def forward(ctx, xp, h0, c0, masks, w_hh, own_xp=False):
        import ctypes as C
        from dcd_hip import check, lib, ptr, stream_ptr
        T, N, G = xp.shape
        H = G // 4
        xp, masks, c0 = xp.contiguous(), masks.contiguous().float(), c0.contiguous()
        # gates accumulate in place (beta = 1): addmm(xp[t], ..., out=act[t]) copied xp[t] per step.  own_xp: the caller formed xp
        # for this call only (RNN._lstm_cuda) and nothing saves it for a backward pass (a Linear's backward needs its input, not
        # its output), so the gates overwrite it instead of a copy — 4.3 GB read + written per PPO epoch at T = 256, N = 4096
        act = xp if own_xp else xp.clone()
        cs = torch.empty((T, N, H), device=xp.device)
        out = torch.empty((T, N, H), device=xp.device)
        hm = torch.empty((T, N, H), device=xp.device)
        torch.mul(h0, masks[0].unsqueeze(1), out=hm[0])
        w_hh_t = w_hh.t()
        L, st = lib(), stream_ptr()
        for t in range(T):
            act[t].addmm_(hm[t], w_hh_t)
            nxt = t + 1 < T
            check(L.dcd_lstm_seq_fwd_step(ptr(act[t]), ptr(cs[t - 1] if t else c0), ptr(masks[t]), ptr(masks[t + 1]) if nxt else None,
                                          ptr(cs[t]), ptr(out[t]), ptr(hm[t + 1]) if nxt else None, C.c_int32(N), C.c_int32(H), st))
        ctx.save_for_backward(act, cs, hm, c0, masks, w_hh)
        return out, out[T - 1].clone(), cs[T - 1].clone()

    @staticmethod
    def backward(ctx, d_out, d_hT, d_cT):
        import ctypes as C
        from dcd_hip import check, lib, ptr, stream_ptr
        act, cs, hm, c0, masks, w_hh = ctx.saved_tensors
        T, N, G = act.shape
        H = G // 4
        dg = torch.empty_like(act)
        dhm = torch.empty((N, H), device=act.device)
        dcf = [torch.empty((N, H), device=act.device) for _ in range(2)]
        d_out = d_out.contiguous() if d_out is not None else None
        d_hT = d_hT.contiguous() if d_hT is not None else None
        d_cT = d_cT.contiguous() if d_cT is not None else None
        L, st = lib(), stream_ptr()
        # dhm = dg[t] @ W_hh with the weight read through a transposed view of a contiguous [H, 4H] copy: hipBLASLt picks a
        # 6 % faster kernel for that operand layout (19.9 vs 21.2 us at N = 4096, tools/lstm_gemm_probe.py; the forward product
        # is fastest as it is) — 1 MB copied once per backward pass, 1 280 GEMMs per student update
        w_hh_v = w_hh.t().contiguous().t()
        for t in range(T - 1, -1, -1):
            last = t == T - 1
            cur, prev = dcf[t & 1], dcf[(t + 1) & 1]
            check(L.dcd_lstm_seq_bwd_step(ptr(act[t]), ptr(cs[t]), ptr(cs[t - 1] if t else c0), ptr(masks[t]),
                                          ptr(d_out[t]) if d_out is not None else None,
                                          None if last else ptr(dhm), None if last else ptr(prev), None if last else ptr(masks[t + 1]),
                                          ptr(d_hT) if (last and d_hT is not None) else None,
                                          ptr(d_cT) if (last and d_cT is not None) else None,
                                          ptr(dg[t]), ptr(cur), C.c_int32(N), C.c_int32(H), st))
            torch.mm(dg[t], w_hh_v, out=dhm)
        m0 = masks[0].unsqueeze(1)
        d_h0 = dhm * m0 if ctx.needs_input_grad[1] else None
        d_c0 = dcf[0] * m0 if ctx.needs_input_grad[2] else None
        d_w = matmul_tn(dg.view(T * N, G), hm.view(T * N, H)) if ctx.needs_input_grad[4] else None
        return dg, d_h0, d_c0, None, d_w, None


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

    def _lstm_cuda(self, x, hxs, masks, single_step):
        """Masked LSTM with explicit GEMMs + pointwise ops (no MIOpen RNN: see _valid_conv_gemm).  The state is
        multiplied by masks[t] before EVERY step, which equals the reference's chunking (inside a chunk the
        masks are all ones, common.py:157-207); the input projection of all T steps is one GEMM."""
        r = self.rnn
        H = self._hidden_size
        h, c = hxs
        if single_step:
            gates = torch.addmm(r.bias_ih_l0 + r.bias_hh_l0, x, r.weight_ih_l0.t()) + (h * masks) @ r.weight_hh_l0.t()
            i, f, g, o = gates.split(H, dim=1)
            c = torch.sigmoid(f) * (c * masks) + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            return h, (h, c)
        N = h.size(0)
        T = x.size(0) // N
        xp = linear(x, r.weight_ih_l0, r.bias_ih_l0 + r.bias_hh_l0)
        return self.forward_projected(xp, hxs, masks, own_xp=True)

    def forward_projected(self, xp, hxs, masks, own_xp=False):
        """The sequence recurrence on an input projection `xp` [T*N, 4H] = x @ W_ih^T + b_ih + b_hh that the caller
        has already formed (MultigridNetwork forms the teacher's incrementally, models/multigrid_models.py)."""
        H = self._hidden_size
        h, c = hxs
        N = h.size(0)
        T = xp.size(0) // N
        xp = xp.view(T, N, 4 * H)
        if xp.is_cuda and _fused_lstm_enabled():
            out, hT, cT = MaskedLSTMSequence.apply(xp, h, c, masks.view(T, N), self.rnn.weight_hh_l0, own_xp)
            return out.view(T * N, H), (hT, cT)
        return self._lstm_cuda_unfused(xp, h, c, masks, T, N)

    def _lstm_cuda_unfused(self, xp, h, c, masks, T, N):
        """The same recurrence composed from PyTorch ops (~40 launches per step with autograd); kept as the
        reference the fused path is tested against (DCD_FUSED_LSTM=0 selects it)."""
        H = self._hidden_size
        # unbind (ONE UnbindBackward that stacks the per-step grads) — indexing xp[t] in the loop would make
        # every step's backward materialise a full [T,N,4H] zero tensor (4.3 GB at T=256, N=4096)
        xp = xp.unbind(0)
        m = masks.view(T, N, 1).unbind(0)
        w_hh_t = self.rnn.weight_hh_l0.t()
        outs = []
        for t in range(T):
            h, c = h * m[t], c * m[t]
            i, f, g, o = (xp[t] + h @ w_hh_t).split(H, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        return torch.stack(outs, 0).view(T * N, H), (h, c)

    def forward(self, x, hxs, masks):
        n_state = hxs[0].size(0) if self.is_lstm else hxs.size(0)
        if self.is_lstm and x.is_cuda:
            return self._lstm_cuda(x, hxs, masks, x.size(0) == n_state)
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
        layers += [init_tanh_(Linear(input_size if i == 0 else sizes[0], sizes[i + 1])), nn.Tanh()]
    return nn.Sequential(*layers)
