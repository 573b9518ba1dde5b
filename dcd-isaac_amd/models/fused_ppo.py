"""The student network's `evaluate_actions` on the PPO batch as fused HIP kernels around four GEMMs (csrc/ppo_kernels.hip).

Reference semantics: models/multigrid_models.py:109-139 (base forward), :169-186 (`evaluate_actions`),
models/distributions.py:18-56 (FixedCategorical), applied to T*B = 1 048 576 rows per epoch by algos/ppo.py:83-116.
Per epoch this replaces ~90 PyTorch launches (im2col stack + transposing copies, `cat`, per-head GEMMs and their dx + add,
row-sums of narrow gradients, ~30 elementwise kernels) by

  features  : dcd_ppo_features_fwd          x1 = [relu(conv) | embed | 1 | 0 0]            [R, 152]
  projection: ONE GEMM                      xp = x1 @ [W_ih | b_ih + b_hh | 0 0]^T         (the bias is a column: no bias pass,
                                                                                             and d bias is a column of dW)
  recurrence: MaskedLSTMSequence            (models/common.py; unchanged)
  heads     : ONE GEMM + dcd_ppo_heads_fwd  pre = h @ [actor.0.0.weight ; critic.0.0.weight]^T  [R, 64] -> value, log_prob, entropy
  backward  : dcd_ppo_heads_bwd -> 2 GEMMs (dh, d first-layer weights) -> LSTM backward -> 2 GEMMs (dx1, d[W_ih | b]) ->
              dcd_ppo_features_bwd (d conv / d embedding as per-workgroup partial sums, added in a fixed order)

Same parameters, same state dict, same arithmetic up to fp32 summation order (tests/test_gpu_fused_ppo.py: every output and
every gradient against the composed PyTorch ops; tests/test_gpu_ppo_vs_reference.py: PPO.update against the reference's numbers).
`DCD_FUSED_PPO=0` selects the composed path.
"""
import ctypes as C
import os

import torch

from . import common

KP = 152                 # operand pitch: 144 conv features + 5 embedding + the constant 1, padded to a multiple of 8


def enabled():
    return os.environ.get("DCD_FUSED_PPO", "1") != "0"


def supported(net):
    """The registered student (util/make_agent.py:38-47): 5 x 5 view, Conv(3 -> 16), Linear(one_hot 4 -> 5), LSTM 256, heads
    256 -> 32 -> {<= 8 actions, 1}."""
    try:
        conv = net.image_conv[0]
        if net.random or net.rnn is None or not net.rnn.is_lstm or net.scalar_embed is None or net.xy_dim:
            return False
        if conv.kernel_size != (3, 3) or conv.in_channels != 3 or conv.out_channels != 16 or net.image_embedding_size != 144:
            return False
        if net.preprocessed_input_size != 144 + net.scalar_embed.out_features or net.scalar_embed.out_features > 7 or net.scalar_embed.in_features > 8:
            return False
        if len(net.actor[0]) != 2 or len(net.critic[0]) != 2:
            return False
        a1, c1, a2, c2 = net.actor[0][0], net.critic[0][0], net.actor[1].linear, net.critic[1]
        return (a1.out_features == 32 and c1.out_features == 32 and a2.out_features <= 8 and c2.out_features == 1
                and a1.in_features == c1.in_features == net.rnn.rnn.hidden_size)
    except (AttributeError, IndexError, TypeError):
        return False


def _struct(net):
    """DcdStudentNet over the network's CURRENT parameter storage (pointers only; rebuilt per call: it is 12 pointers)."""
    from dcd_hip import ptr
    from dcd_hip.policy import StudentNet
    conv, emb = net.image_conv[0], net.scalar_embed
    a1, a2, c1, c2 = net.actor[0][0], net.actor[1].linear, net.critic[0][0], net.critic[1]
    s = StudentNet()
    params = [conv.weight, conv.bias, emb.weight, emb.bias, a1.weight, a1.bias, a2.weight, a2.bias, c1.weight, c1.bias, c2.weight, c2.bias]
    for name, p in zip([f[0] for f in StudentNet._fields_[:12]], params):
        setattr(s, name, ptr(p.detach(), torch.float32, name=name).value)
    s.V, s.F, s.E, s.D, s.H = 5, 16, emb.out_features, emb.in_features, net.rnn.rnn.hidden_size
    s.A1, s.C1, s.NA = a1.out_features, c1.out_features, a2.out_features
    return s


def _rows_per_wg(R, unit, target=1024):
    """Rows per workgroup of a backward kernel: ~1024 workgroups, a multiple of the kernel's tile."""
    per = (R + target - 1) // target
    return max(unit, (per + unit - 1) // unit * unit)


class StudentFeatures(torch.autograd.Function):
    """x1 [R, 152] from (image, direction): fp32 CHW image + float direction, or the raw bytes (u8 [R,5,5,3] + u8 [R])."""

    @staticmethod
    def forward(ctx, net, image, direction, conv_w, conv_b, emb_w, emb_b):
        from dcd_hip import check, lib, ptr, stream_ptr
        u8 = image.dtype == torch.uint8
        R = image.shape[0]
        image, direction = image.contiguous(), direction.contiguous().view(-1)
        assert direction.dtype == (torch.uint8 if u8 else torch.float32) and direction.numel() == R and image.numel() == R * 75
        x = torch.empty((R, KP), device=image.device, dtype=torch.float32)
        s = _struct(net)
        check(lib().dcd_ppo_features_fwd(C.byref(s), ptr(image), ptr(direction), C.c_int32(int(u8)), ptr(x), C.c_int32(KP), C.c_int64(R), stream_ptr()))
        ctx.net, ctx.u8 = net, u8
        ctx.save_for_backward(image, direction, x)
        return x

    @staticmethod
    def backward(ctx, dx):
        from dcd_hip import check, lib, ptr, stream_ptr
        image, direction, x = ctx.saved_tensors
        net, R = ctx.net, x.shape[0]
        dx = dx.contiguous()
        rpw = _rows_per_wg(R, 16)
        nwg = (R + rpw - 1) // rpw
        FP = lib().dcd_ppo_features_partial_floats()
        part = torch.empty((nwg, FP), device=x.device, dtype=torch.float32)
        s = _struct(net)
        check(lib().dcd_ppo_features_bwd(C.byref(s), ptr(image), ptr(direction), C.c_int32(int(ctx.u8)), ptr(x), ptr(dx), C.c_int32(KP), C.c_int64(R),
                                         C.c_int32(rpw), ptr(part), stream_ptr()))
        g = part.sum(0)
        E, D = s.E, s.D
        d_emb_w = g[448:512].view(8, 8)[:E, :D].contiguous()
        return None, None, None, g[:432].view(16, 3, 3, 3), g[432:448], d_emb_w, d_emb_w.sum(1)


class LinearOnes(torch.autograd.Function):
    """xp = x1 @ [W | b | 0..]^T for an operand x1 whose column W.shape[1] is the constant 1 (and whose later columns are 0):
    d W and d b come out of ONE split-K contraction, d x1 of one GEMM; no bias broadcast, no row-sum."""

    @staticmethod
    def forward(ctx, x1, w, b):
        K = w.shape[1]
        w1 = torch.zeros((w.shape[0], x1.shape[1]), device=w.device, dtype=w.dtype)
        w1[:, :K] = w
        w1[:, K] = b
        ctx.save_for_backward(x1, w1)
        ctx.K = K
        return x1 @ w1.t()

    @staticmethod
    def backward(ctx, g):
        x1, w1 = ctx.saved_tensors
        K = ctx.K
        g = g.contiguous()
        dx1 = g @ w1 if ctx.needs_input_grad[0] else None
        dw1 = common.matmul_tn(g, x1)
        return dx1, dw1[:, :K], dw1[:, K]


class StudentHeads(torch.autograd.Function):
    """(value [R,1], log_prob(action) [R,1], entropy per row [R]) from the LSTM outputs h [R, H]."""

    @staticmethod
    def forward(ctx, net, h, actions, a1w, a1b, a2w, a2b, c1w, c1b, c2w, c2b):
        from dcd_hip import check, lib, ptr, stream_ptr
        R = h.shape[0]
        h = h.contiguous()
        actions = actions.contiguous().view(-1)
        assert actions.dtype == torch.int64 and actions.numel() == R
        w1 = torch.cat([a1w, c1w], 0)                                    # [64, H]: both first layers in one GEMM
        act = h @ w1.t()                                                 # pre-activations; the kernel turns them into tanh outputs
        logp_all = torch.empty((R, 8), device=h.device, dtype=torch.float32)
        value = torch.empty((R, 1), device=h.device, dtype=torch.float32)
        logp = torch.empty((R, 1), device=h.device, dtype=torch.float32)
        ent = torch.empty((R,), device=h.device, dtype=torch.float32)
        s = _struct(net)
        check(lib().dcd_ppo_heads_fwd(C.byref(s), ptr(act), ptr(actions), ptr(logp_all), ptr(value), ptr(logp), ptr(ent), C.c_int64(R), stream_ptr()))
        ctx.net = net
        ctx.save_for_backward(h, actions, act, logp_all, w1)
        return value, logp, ent

    @staticmethod
    def backward(ctx, dvalue, dlogp, dent):
        from dcd_hip import check, lib, ptr, stream_ptr
        h, actions, act, logp_all, w1 = ctx.saved_tensors
        net, R = ctx.net, h.shape[0]
        f = lambda t: None if t is None else t.contiguous().view(-1)
        dvalue, dlogp, dent = f(dvalue), f(dlogp), f(dent)
        rpw = _rows_per_wg(R, 128)
        nwg = (R + rpw - 1) // rpw
        HP = lib().dcd_ppo_heads_partial_floats()
        part = torch.empty((nwg, HP), device=h.device, dtype=torch.float32)
        s = _struct(net)
        # `act` is overwritten by the d pre-activations: it is this node's own buffer and the node runs its backward once
        check(lib().dcd_ppo_heads_bwd(C.byref(s), ptr(act), ptr(logp_all), ptr(actions), ptr(dvalue), ptr(dlogp), ptr(dent), C.c_int64(R),
                                      C.c_int32(rpw), ptr(part), stream_ptr()))
        g = part.sum(0)
        NA = s.NA
        dpre = act
        dh = dpre @ w1 if ctx.needs_input_grad[1] else None
        dw1 = common.matmul_tn(dpre, h)                                         # [64, H]
        return (None, dh, None, dw1[:32], g[304:336], g[:256].view(8, 32)[:NA].contiguous(), g[256:256 + NA], dw1[32:], g[336:368],
                g[264:296].view(1, 32), g[296:297])


def evaluate_actions(net, inputs, rnn_hxs, masks, action):
    """MultigridNetwork.evaluate_actions (multigrid_models.py:169-186) for a `supported` network on the device."""
    conv, emb, lstm = net.image_conv[0], net.scalar_embed, net.rnn.rnn
    if "image_u8" in inputs:
        image, direction = inputs["image_u8"], inputs["direction_u8"]
    else:
        image, direction = inputs["image"], inputs["direction"]
    x1 = StudentFeatures.apply(net, image, direction, conv.weight, conv.bias, emb.weight, emb.bias)
    xp = LinearOnes.apply(x1, lstm.weight_ih_l0, lstm.bias_ih_l0 + lstm.bias_hh_l0)
    out, rnn_hxs = net.rnn.forward_projected(xp, rnn_hxs, masks, own_xp=True)
    a1, a2, c1, c2 = net.actor[0][0], net.actor[1].linear, net.critic[0][0], net.critic[1]
    value, logp, ent = StudentHeads.apply(net, out, action, a1.weight, a1.bias, a2.weight, a2.bias, c1.weight, c1.bias, c2.weight, c2.bias)
    return value, logp, ent.mean(), rnn_hxs
