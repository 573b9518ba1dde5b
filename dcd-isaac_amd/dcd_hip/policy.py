"""Rollout-time student policy step on the HIP path (dcd_policy_* of include/dcd_hip.h).

`StudentActor` wraps a `MultigridNetwork` (student shape: image + direction, one conv, LSTM-256, one tanh
layer per head) and performs `act` + `storage.insert_policy` for a whole env batch with two HIP kernels and
one rocBLAS GEMM, writing straight into the RolloutStorage slices — the eager PyTorch path costs ~45 launches
per step and is host-bound (0.63 ms per step at B = 4096; this path: see DESIGN.md §5).

The network's parameters stay ordinary PyTorch parameters (PPO trains them through the PyTorch modules); this
class only reads them.  `try_create` returns None when the network is not of the supported shape — callers
then use the PyTorch modules on the GPU (same results, more launches); there is no CPU path.
"""
import ctypes as C
import os

import torch

from . import DcdHipError, check, lib, ptr, stream_ptr


class StudentNet(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("conv_w", "conv_b", "embed_w", "embed_b", "actor_w1", "actor_b1", "actor_w2",
                                          "actor_b2", "critic_w1", "critic_b1", "critic_w2", "critic_b2")] + \
               [(k, C.c_int32) for k in ("V", "F", "E", "D", "H", "A1", "C1", "NA")]


class PolicyAgent(C.Structure):
    """DcdPolicyAgent (include/dcd_hip.h): one student's pointers for the two-students-per-launch entry points."""
    _fields_ = [(k, C.c_void_p) for k in ("net", "packed", "image_u8", "dir_u8", "hc_in", "masks", "x", "gates", "hc_out", "value",
                                          "action", "log_prob", "log_dist", "u_out", "counter_base")] + \
               [("env_index_base", C.c_int64), ("seed", C.c_uint32), ("counter", C.c_uint32), ("deterministic", C.c_int32),
                ("reserved", C.c_int32)]


def enabled():
    return os.environ.get("DCD_FUSED_ACT", "1") != "0"


class StudentActor:
    def __init__(self, net, num_envs, seed=0, env_index_base=0, buffers=None):
        """buffers: optional (x, gates, w_t, bias) tensors to use instead of own allocations — `PairedStudentActors` hands
        each of two actors one half of [2, ...] tensors so that both GEMMs of a step are one batched GEMM."""
        self.net, self.B = net, int(num_envs)
        self.seed, self.env_index_base = int(seed) & 0xFFFFFFFF, int(env_index_base)
        self.counter = 0                                            # immediate part of the Philox counter (step index)
        conv = net.image_conv[0]
        lstm = net.rnn.rnn
        self.H = lstm.hidden_size
        dev = conv.weight.device
        self.device = dev
        V = int(round((net.image_embedding_size / conv.out_channels) ** 0.5)) + conv.kernel_size[0] - 1
        a1, a2 = net.actor[0][0], net.actor[1].linear
        c1, c2 = net.critic[0][0], net.critic[1]
        self._params = [conv.weight, conv.bias, net.scalar_embed.weight, net.scalar_embed.bias, a1.weight, a1.bias,
                        a2.weight, a2.bias, c1.weight, c1.bias, c2.weight, c2.bias]
        self._dims = dict(V=V, F=conv.out_channels, E=net.scalar_embed.out_features, D=net.scalar_embed.in_features,
                          H=self.H, A1=a1.out_features, C1=c1.out_features, NA=a2.out_features)
        self._lstm = lstm
        self.K = self._dims["F"] * (V - 2) ** 2 + self._dims["E"] + self.H
        # K padded for the GEMM (405 -> 416) with at least one spare column: column K of x is the constant 1 and row K of the
        # stacked weight is b_ih + b_hh, i.e. the bias is one more term of every dot product and the GEMM needs no bias
        # operand (a batched GEMM with a broadcast bias costs an extra 33 MB copy per step); the other pad columns are zero
        self.KP = (self.K + 1 + 31) // 32 * 32
        if buffers is not None:
            self.x, self.gates, self.w_t, self.bias = buffers
            assert self.x.shape == (self.B, self.KP) and self.w_t.shape == (self.KP, 4 * self.H) and self.x.is_contiguous()
        else:
            self.x = torch.zeros((self.B, self.KP), device=dev)
            self.gates = torch.empty((self.B, 4 * self.H), device=dev)
            self.w_t = torch.zeros((self.KP, 4 * self.H), device=dev)     # persistent: captured HIP graphs keep pointing at them
            self.bias = torch.empty((4 * self.H,), device=dev)
        self.x[:, self.K] = 1.0
        self.packed = torch.empty((lib().dcd_policy_packed_heads_floats(),), device=dev)   # head weights in the kernel's layout
        self.counter_base = torch.zeros(1, dtype=torch.int32, device=dev)   # device part of the counter (rollouts so far x T)
        self._struct = None
        self._w_version = None
        self.refresh()

    # ---- construction ------------------------------------------------------------------------
    @staticmethod
    def supported(net):
        try:
            from models.multigrid_models import MultigridNetwork
        except ImportError:
            return False
        if not isinstance(net, MultigridNetwork) or net.random or net.rnn is None or not net.rnn.is_lstm:
            return False
        conv = net.image_conv[0]
        if not conv.weight.is_cuda or conv.kernel_size != (3, 3) or conv.in_channels != 3 or conv.out_channels > 32:
            return False
        if net.scalar_embed is None or net.xy_dim or net.preprocessed_input_size != net.image_embedding_size + net.scalar_embed.out_features:
            return False                                            # random_z / xy inputs: teacher-shaped nets
        if net.rnn.rnn.hidden_size != 256 or net.rnn.rnn.num_layers != 1:
            return False
        if len(net.actor[0]) != 2 or len(net.critic[0]) != 2:        # exactly one Linear+Tanh per head
            return False
        a1, c1, a2 = net.actor[0][0], net.critic[0][0], net.actor[1].linear
        return a1.out_features + c1.out_features <= 64 and a2.out_features < 16 and net.critic[1].out_features == 1

    @classmethod
    def try_create(cls, net, num_envs, seed=0, env_index_base=0, buffers=None):
        if not enabled() or not cls.supported(net):
            return None
        return cls(net, num_envs, seed=seed, env_index_base=env_index_base, buffers=buffers)

    @staticmethod
    def input_pitch(net):
        """KP of a supported network (the GEMM operand's row pitch): known before the actor exists."""
        k = net.image_embedding_size + net.scalar_embed.out_features + net.rnn.rnn.hidden_size
        return (k + 1 + 31) // 32 * 32

    def agent_struct(self, st, step, deterministic=False):
        """DcdPolicyAgent for `act_storage(st, step)` (u8 rollout buffers)."""
        B, H, V, NA = self.B, self.H, self._dims["V"], self._dims["NA"]
        o = st.obs_u8
        a = PolicyAgent()
        self._struct_keep = self._struct
        a.net = C.addressof(self._struct)
        a.packed = ptr(self.packed).value
        a.image_u8 = ptr(o["image"][step], torch.uint8, B * 3 * V * V, "image_u8").value
        a.dir_u8 = ptr(o["direction"][step], torch.uint8, B, "dir_u8").value
        a.hc_in = ptr(st.recurrent_hidden_states[step], torch.float32, B * 2 * H, "hc_in").value
        a.masks = ptr(st.masks[step], torch.float32, B, "masks").value
        a.x, a.gates = ptr(self.x).value, ptr(self.gates).value
        a.hc_out = ptr(st.recurrent_hidden_states[step + 1], torch.float32, B * 2 * H, "hc_out").value
        a.value = ptr(st.value_preds[step], torch.float32, B, "value").value
        a.action = ptr(st.actions[step], torch.int64, B, "action").value
        a.log_prob = ptr(st.action_log_probs[step], torch.float32, B, "log_prob").value
        a.log_dist = ptr(st.action_log_dist[step], torch.float32, B * NA, "log_dist").value
        a.u_out = None
        a.counter_base = self.counter_base.data_ptr()
        a.env_index_base = self.env_index_base
        a.seed, a.counter, a.deterministic = self.seed, step & 0xFFFFFFFF, int(deterministic)
        return a

    def refresh(self):
        """Re-read parameter pointers and rebuild, IN PLACE, the two derived buffers — the stacked LSTM weight
        [K, 4H] and the packed head weights — when the optimiser has stepped (in-place updates keep pointers
        but move the parameters' version counters)."""
        l = self._lstm
        ver = tuple(p._version for p in (l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0)) + \
            tuple((p.data_ptr(), p._version) for p in self._params)
        if ver == self._w_version:
            return
        with torch.no_grad():
            self.w_t[:self.K - self.H].copy_(l.weight_ih_l0.t())       # [K, 4H] = [W_ih | W_hh]^T, refreshed in place
            self.w_t[self.K - self.H:self.K].copy_(l.weight_hh_l0.t())
            torch.add(l.bias_ih_l0, l.bias_hh_l0, out=self.bias)
            self.w_t[self.K].copy_(self.bias)
        s = StudentNet()
        for name, p in zip([f[0] for f in StudentNet._fields_[:12]], self._params):
            setattr(s, name, ptr(p.detach(), torch.float32, name=name).value)
        for k, v in self._dims.items():
            setattr(s, k, v)
        self._struct = s
        self._w_version = ver
        check(lib().dcd_policy_pack_heads(C.byref(s), ptr(self.packed), stream_ptr()))
        if lib().dcd_policy_input_width(C.byref(s)) != self.K:
            raise DcdHipError("dcd_policy_input_width disagrees with the network's LSTM input size")

    # ---- one step ----------------------------------------------------------------------------
    def act_into(self, image, direction, hc_in, masks, hc_out, value, action, log_prob, log_dist, deterministic=False,
                 u_out=None):
        """All arguments are slices of a RolloutStorage (contiguous CUDA tensors):
        image [B,3,V,V], direction [B,1], hc_in/hc_out [B,2H], masks [B,1] -> value [B,1], action int64 [B,1],
        log_prob [B,1], log_dist [B,NA]."""
        B, s = self.B, self._struct
        L = lib()
        if image.dtype == torch.uint8:                                # raw observation bytes (RolloutStorage in u8 mode)
            check(L.dcd_policy_features_u8(C.byref(s), ptr(image, torch.uint8, B * 3 * self._dims["V"] ** 2, "image_u8"),
                                           ptr(direction, torch.uint8, B, "dir_u8"), ptr(hc_in, torch.float32, B * 2 * self.H, "hc_in"),
                                           ptr(masks, torch.float32, B, "masks"), ptr(self.x), C.c_int32(self.KP), C.c_int32(B), stream_ptr()))
        else:
            check(L.dcd_policy_features(C.byref(s), ptr(image, torch.float32, B * 3 * self._dims["V"] ** 2, "image"),
                                        ptr(direction, torch.float32, B, "direction"), ptr(hc_in, torch.float32, B * 2 * self.H, "hc_in"),
                                        ptr(masks, torch.float32, B, "masks"), ptr(self.x), C.c_int32(self.KP), C.c_int32(B), stream_ptr()))
        torch.mm(self.x, self.w_t, out=self.gates)
        check(L.dcd_policy_cell_heads(C.byref(s), ptr(self.packed), ptr(self.gates), ptr(hc_in), ptr(masks), ptr(hc_out, torch.float32, B * 2 * self.H, "hc_out"),
                                      ptr(value, torch.float32, B, "value"), ptr(action, torch.int64, B, "action"),
                                      ptr(log_prob, torch.float32, B, "log_prob"),
                                      ptr(log_dist, torch.float32, B * self._dims["NA"], "log_dist"),
                                      ptr(u_out, torch.float32, B, "u_out") if u_out is not None else None,
                                      C.c_uint32(self.seed), C.c_uint32(self.counter & 0xFFFFFFFF),
                                      C.c_void_p(self.counter_base.data_ptr()), C.c_int64(self.env_index_base),
                                      C.c_int32(int(deterministic)), C.c_int32(B), stream_ptr()))
        self.counter += 1

    def begin_rollout(self, num_steps):
        """Call once per rollout, OUTSIDE any graph capture: picks up optimiser steps and advances the device
        counter base by the rollout length, so step t of this rollout draws with counter base + t."""
        self.refresh()
        self.counter_base.add_(int(num_steps))

    def act_storage(self, st, step, deterministic=False):
        """`agent.act(obs[step], h[step], masks[step])` + `storage.insert_policy(...)` (adversarial_runner.py:506-517,
        algos/storage.py:155-185) without leaving the device: writes value_preds[step], actions[step],
        action_log_probs[step], action_log_dist[step], recurrent_hidden_states[step+1]."""
        self.counter = step
        o = st.obs_u8 if st.obs_u8 is not None else st._obs
        self.act_into(o["image"][step], o["direction"][step], st.recurrent_hidden_states[step], st.masks[step],
                      st.recurrent_hidden_states[step + 1], st.value_preds[step], st.actions[step], st.action_log_probs[step],
                      st.action_log_dist[step], deterministic=deterministic)

    def value_storage(self, st, step):
        """`agent.get_value(obs[step], h[step], masks[step])` (adversarial_runner.py:586-590: the bootstrap value of a
        finished rollout) by the same three launches as a rollout step, written into value_preds[step]; the sampled
        action and the next hidden state go to scratch.  Same value as the step kernels give during the rollout."""
        if getattr(self, "_scratch", None) is None:
            B, dev = self.B, self.device
            self._scratch = (torch.empty((B, 2 * self.H), device=dev), torch.empty((B, 1), dtype=torch.int64, device=dev),
                             torch.empty((B, 1), device=dev), torch.empty((B, self._dims["NA"]), device=dev))
        hc, action, log_prob, log_dist = self._scratch
        self.refresh()
        self.counter = step
        o = st.obs_u8 if st.obs_u8 is not None else st._obs
        self.act_into(o["image"][step], o["direction"][step], st.recurrent_hidden_states[step], st.masks[step],
                      hc, st.value_preds[step], action, log_prob, log_dist, deterministic=True)
        return st.value_preds[step]


class PairedStudentActors:
    """Two StudentActors of one geometry (PAIRED's protagonist and antagonist) acting at the same step of their lock-step
    rollouts: ONE features launch, ONE batched GEMM, ONE cell + heads launch for both (dcd_policy_features_u8_pair,
    dcd_policy_cell_heads_pair).  The actors must have been created on the halves of `buffers(...)`."""

    @staticmethod
    def buffers(nets, num_envs):
        """[2, ...] operand / gate / weight / bias tensors for two supported networks, or None if their shapes differ."""
        kp = [StudentActor.input_pitch(n) for n in nets]
        hs = [n.rnn.rnn.hidden_size for n in nets]
        if len(nets) != 2 or kp[0] != kp[1] or hs[0] != hs[1]:
            return None
        dev, B, G = nets[0].image_conv[0].weight.device, int(num_envs), 4 * hs[0]
        return dict(x=torch.zeros((2, B, kp[0]), device=dev), gates=torch.empty((2, B, G), device=dev),
                    w_t=torch.zeros((2, kp[0], G), device=dev), bias=torch.empty((2, 1, G), device=dev))

    @staticmethod
    def half(buf, k):
        return buf["x"][k], buf["gates"][k], buf["w_t"][k], buf["bias"][k, 0]

    def __init__(self, actors, buf):
        self.actors, self.buf = tuple(actors), buf
        for k, a in enumerate(self.actors):
            assert a.x.data_ptr() == buf["x"][k].data_ptr() and a.w_t.data_ptr() == buf["w_t"][k].data_ptr()

    def act_storage(self, sts, step, deterministic=False):
        L, B = lib(), self.actors[0].B
        a0, a1 = (f.agent_struct(st, step, deterministic) for f, st in zip(self.actors, sts))
        check(L.dcd_policy_features_u8_pair(C.byref(a0), C.byref(a1), C.c_int32(self.actors[0].KP), C.c_int32(B), stream_ptr()))
        torch.bmm(self.buf["x"], self.buf["w_t"], out=self.buf["gates"])
        check(L.dcd_policy_cell_heads_pair(C.byref(a0), C.byref(a1), C.c_int32(B), stream_ptr()))
        for f in self.actors:
            f.counter = step + 1


_DIV10 = {}


def _div10_lut(device):
    """k -> fp32(k / 10.0) for every byte: VecPreprocessImageWrapper's scale (obs_wrappers.py:90-91), exact."""
    key = str(device)
    if key not in _DIV10:
        import numpy as np
        _DIV10[key] = torch.from_numpy((np.arange(256, dtype=np.float64) / 10.0).astype(np.float32)).to(device)
    return _DIV10[key]


def _u8_to_chw(img8):
    """[n,S,S,3] raw encoding -> the preprocessed [n,3,S,S] float image (small inputs only: first frames)."""
    return _div10_lut(img8.device)[img8.long()].permute(0, 3, 1, 2).contiguous()


def incremental_act_enabled():
    """DCD_INCREMENTAL_ACT=0: the teacher's rollout multiplies the dense 21 632-wide operand at every step."""
    return os.environ.get("DCD_INCREMENTAL_ACT", "1") != "0"


class GridActor:
    """Rollout-time policy step of a recurrent, grid-observing MultigridNetwork (the PAIRED teacher: 15 x 15 image,
    128 filters, time-step embedding, random_z, LSTM-256): `dcd_policy_grid_features` writes the GEMM operand row
    [relu(conv) | embed | z | h*mask] directly — the module path builds it through an im2col copy chain and a 355 MB
    `cat` per step (0.9 of its 2.4 ms) — then ONE GEMM against the stacked [W_ih | W_hh]; the cell and the heads stay
    PyTorch ops on [B, 256] tensors (169-way categorical: `torch.multinomial`, as in the reference)."""

    def __init__(self, net, num_envs):
        self.net, self.B = net, int(num_envs)
        conv, lstm = net.image_conv[0], net.rnn.rnn
        self.H, self._lstm = lstm.hidden_size, lstm
        dev = conv.weight.device
        F = conv.out_channels
        P = net.image_embedding_size // F
        S = int(round(P ** 0.5)) + 2
        E, D = net.scalar_embed.out_features, net.scalar_embed.in_features
        self.Z = net.preprocessed_input_size - net.image_embedding_size - E
        self._dims = dict(V=S, F=F, E=E, D=D, H=self.H, A1=0, C1=0, NA=0)
        self._params = [conv.weight, conv.bias, net.scalar_embed.weight, net.scalar_embed.bias]
        self.K = F * P + E + self.Z + self.H
        self.KP = (self.K + 31) // 32 * 32
        self.x = torch.zeros((self.B, self.KP), device=dev)
        self.gates = torch.empty((self.B, 4 * self.H), device=dev)
        self.w_t = torch.zeros((self.KP, 4 * self.H), device=dev)
        self.bias = torch.empty((4 * self.H,), device=dev)
        # fused step (`act_storage`): heads of one Linear + Tanh each, < 192 actions, H = 256 -> everything after the carried
        # projection is two HIP kernels around one GEMM and lands in the storage; else `act` returns tensors as the module does
        self.fused_heads = self.heads_supported(net)
        self.seed, self.env_index_base = 0, 0
        if self.fused_heads:
            a1, a2, c1, c2 = net.actor[0][0], net.actor[1].linear, net.critic[0][0], net.critic[1]
            self._params += [a1.weight, a1.bias, a2.weight, a2.bias, c1.weight, c1.bias, c2.weight, c2.bias]
            self._dims.update(A1=a1.out_features, C1=c1.out_features, NA=a2.out_features)
            self.KR = (E + self.Z + self.H + 1 + 31) // 32 * 32       # operand row of everything but the image (+ the bias column)
            self.x_rest = torch.zeros((self.B, self.KR), device=dev)
            self.w_rest = torch.zeros((self.KR, 4 * self.H), device=dev)
            self.counter_base = torch.zeros(1, dtype=torch.int32, device=dev)
            self.packed = None
        self._struct, self._w_version = None, None
        self.refresh()

    @staticmethod
    def heads_supported(net):
        try:
            if len(net.actor[0]) != 2 or len(net.critic[0]) != 2 or net.rnn.rnn.hidden_size != 256:
                return False
            a1, a2, c1, c2 = net.actor[0][0], net.actor[1].linear, net.critic[0][0], net.critic[1]
            return (a1.out_features <= 32 and c1.out_features <= 32 and a2.out_features < 192 and c2.out_features == 1
                    and a2.in_features == a1.out_features and c2.in_features == c1.out_features)
        except (AttributeError, IndexError, TypeError):
            return False

    @staticmethod
    def supported(net):
        try:
            from models.multigrid_models import MultigridNetwork
        except ImportError:
            return False
        if not isinstance(net, MultigridNetwork) or net.random or net.rnn is None or not net.rnn.is_lstm:
            return False
        conv = net.image_conv[0]
        if not conv.weight.is_cuda or conv.kernel_size != (3, 3) or conv.in_channels != 3 or conv.out_channels > 128:
            return False
        if net.scalar_embed is None or net.xy_dim or net.rnn.rnn.num_layers != 1 or net.rnn.rnn.hidden_size > 256:
            return False
        P = net.image_embedding_size // conv.out_channels
        S = int(round(P ** 0.5)) + 2
        return (S - 2) ** 2 == P and 9 <= S <= 17 and net.scalar_embed.out_features <= 256

    @classmethod
    def try_create(cls, net, num_envs):
        if not enabled() or not cls.supported(net):
            return None
        return cls(net, num_envs)

    def refresh(self):
        l = self._lstm
        ver = tuple(p._version for p in (l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0)) + \
            tuple((p.data_ptr(), p._version) for p in self._params)
        if ver == self._w_version:
            return
        with torch.no_grad():
            self.w_t[:self.K - self.H].copy_(l.weight_ih_l0.t())
            self.w_t[self.K - self.H:self.K].copy_(l.weight_hh_l0.t())
            torch.add(l.bias_ih_l0, l.bias_hh_l0, out=self.bias)
            F, G = self._dims["F"], 4 * self.H
            P = (self._dims["V"] - 2) ** 2
            if getattr(self, "wp", None) is None:                  # persistent: captured graphs keep pointing at it
                self.wp = torch.empty((P, F, G), device=self.w_t.device)
            self.wp.copy_(l.weight_ih_l0[:, :F * P].view(G, F, P).permute(2, 1, 0))             # [P, F, 4H]: per-position blocks
        s = StudentNet()
        names = ("conv_w", "conv_b", "embed_w", "embed_b", "actor_w1", "actor_b1", "actor_w2", "actor_b2", "critic_w1", "critic_b1",
                 "critic_w2", "critic_b2")
        for name, p in zip(names, self._params):
            setattr(s, name, ptr(p.detach(), torch.float32, name=name).value)
        for k, v in self._dims.items():
            setattr(s, k, v)
        self._struct, self._w_version = s, ver
        if self.fused_heads:
            FP = self.K - self.H - self.Z - self._dims["E"]
            n_rest = self.K - FP                                          # E + Z + H rows of the stacked weight
            with torch.no_grad():
                self.w_rest[:n_rest].copy_(self.w_t[FP:self.K])
                self.w_rest[n_rest].copy_(self.bias)                      # multiplied by the operand's constant-1 column
            if self.packed is None:
                self.packed = torch.empty((lib().dcd_policy_wide_packed_heads_floats(C.byref(s)),), device=self.w_t.device)
            check(lib().dcd_policy_wide_pack_heads(C.byref(s), ptr(self.packed), stream_ptr()))

    # ---- incremental acting: the image part of the projection is carried from step to step ------------------------
    def begin_episode(self):
        """Call before the first step of a teacher episode: forget the carried projection."""
        self._xp_img, self._incremental = None, incremental_act_enabled()

    def _carry_projection(self, image, prev_image):
        """self._xp_img [B, 4H] = relu(conv(image)).flatten(1) @ W_ih[:, :F*P]^T, carried: at the first step every env
        shows the same empty grid (one row, broadcast); afterwards a teacher action changes one or two cells, i.e.
        <= 18 of the P conv positions — only their feature changes are multiplied (one batched GEMM over positions,
        rows summed by dcd_gather_sum_rows), the same form PPO.update uses (MultigridNetwork._incremental_projection).
        Returns False when this episode has to use the dense operand (first frames differ)."""
        import torch.nn.functional as Fn
        net, B, G = self.net, self.B, 4 * self.H
        S, F = self._dims["V"], self._dims["F"]
        O = S - 2
        P = O * O
        u8 = image.dtype == torch.uint8              # raw encoding [B,S,S,3] ([x][y][c]) instead of the preprocessed [B,3,S,S]
        if prev_image is None:
            if not bool((image == image[:1]).all()):                               # (host sync)
                return False
            first = _u8_to_chw(image[:1]) if u8 else image[:1]
            self._xp_img = (net.image_conv(first) @ self._lstm.weight_ih_l0[:, :F * P].t()).expand(B, G).contiguous()
            return True
        if self._xp_img is None:
            return False
        dev = image.device
        changed = (image != prev_image).any(dim=-1 if u8 else 1)                   # [B,S,S] over (x, y) either way
        dirty = Fn.max_pool2d(changed.float().unsqueeze(1), 3, 1).squeeze(1) > 0   # [B,O,O]: conv positions that see a change
        nn_, oy, ox = dirty.nonzero(as_tuple=True)                                 # (host sync) rows in (env, position) order
        M = nn_.numel()
        if M == 0:
            return True
        p = oy * O + ox
        counts = torch.bincount(p, minlength=P)
        rank = torch.arange(M, device=dev) - torch.searchsorted(nn_, nn_)          # k-th dirty position of its env
        cmax, kmax = torch.stack([counts.max(), rank.max() + 1]).tolist()          # (host sync)
        order = torch.argsort(p, stable=True)
        p_s, n_s = p[order], nn_[order]
        slot = p_s * cmax + (torch.arange(M, device=dev) - (torch.cumsum(counts, 0) - counts)[p_s])
        src = torch.full((B, kmax), -1, dtype=torch.int32, device=dev)
        slot_u = torch.empty_like(slot)
        slot_u[order] = slot
        src[nn_, rank] = slot_u.to(torch.int32)
        k = torch.arange(3, device=dev)
        if u8:                                                                     # element (c, k1, k2) of the window at byte ((y+k1)*S + x+k2)*3 + c
            offs = (torch.arange(3, device=dev).view(3, 1, 1) + (k.view(1, 3, 1) * S + k.view(1, 1, 3)) * 3).reshape(-1)
            col = ((oy[order] * S + ox[order]) * 3).unsqueeze(1) + offs.unsqueeze(0)
        else:
            offs = (torch.arange(3, device=dev).view(3, 1, 1) * S * S + k.view(1, 3, 1) * S + k.view(1, 1, 3)).reshape(-1)
            col = (oy[order] * S + ox[order]).unsqueeze(1) + offs.unsqueeze(0)     # [M, 27]: the 3x3x3 window of each pair
        conv = net.image_conv[0]
        wc = conv.weight.view(F, -1).t()
        rows = n_s.unsqueeze(1)
        window = (lambda im: _div10_lut(dev)[im.reshape(B, -1)[rows, col].long()]) if u8 else (lambda im: im.reshape(B, -1)[rows, col])
        feat = lambda im: torch.relu(torch.addmm(conv.bias, window(im), wc))
        dpad = torch.zeros((P * cmax, F), device=dev).index_copy_(0, slot, feat(image) - feat(prev_image))
        y = torch.bmm(dpad.view(P, cmax, F), self.wp).view(P * cmax, G)
        check(lib().dcd_gather_sum_rows(ptr(y, torch.float32), ptr(src, torch.int32), ptr(self._xp_img, torch.float32), C.c_int32(B),
                                        C.c_int32(kmax), C.c_int32(G), stream_ptr()))
        return True

    # ---- the same, device-resident (raw u8 observations): no host synchronisation, fixed shapes, graph-capturable ---
    def device_carry_supported(self, image):
        return incremental_act_enabled() and image.dtype == torch.uint8 and image.dim() == 4 and image.shape[-1] == 3

    def _device_buffers(self):
        if getattr(self, "_d", None) is None:
            L, dev, B, G, F = lib(), self.x.device, self.B, 4 * self.H, self._dims["F"]
            S = self._dims["V"]
            P = (S - 2) ** 2
            Tt, R = L.dcd_grid_delta_tiles(S, B), L.dcd_grid_delta_tile_rows()
            i32 = dict(dtype=torch.int32, device=dev)
            self._d = dict(Tt=Tt, R=R, cell=torch.zeros(B, **i32), count=torch.zeros(P, **i32), src=torch.full((B, 9), -1, **i32),
                           rows=torch.zeros((Tt * R, F), device=dev), tile=torch.zeros(Tt, **i32),
                           y=torch.zeros((Tt * R, G), device=dev), used=torch.zeros(1, **i32),
                           err=torch.zeros(1, **i32), xp=torch.empty((B, G), device=dev))
        return self._d

    @torch.no_grad()
    def _carry_projection_device(self, image, prev_image):
        """`_carry_projection` for raw u8 images with nothing read back: `dcd_grid_delta_rows` finds each env's changed
        cell, bins the (env, conv position) pairs by position in env order and writes their feature differences as
        128-row tiles; `dcd_grid_tile_gemm` (fp32 MFMA) multiplies each tile that holds rows by its position's weight
        block, found through the device-side tile table; dcd_gather_sum_rows adds an env's <= 9 rows in slot order.  The first
        step takes env 0's image as every env's previous image (after the teacher's reset all grids are the same empty
        room; a grid that differs from it in one cell is still exact).  More than one changed cell in a step sets an error
        word that `check_carry_errors` turns into an exception."""
        d, net, B, G = self._device_buffers(), self.net, self.B, 4 * self.H
        S, F = self._dims["V"], self._dims["F"]
        P = (S - 2) ** 2
        L = lib()
        if prev_image is None:
            xp0 = net.image_conv(_u8_to_chw(image[:1])) @ self._lstm.weight_ih_l0[:, :F * P].t()
            d["xp"].copy_(xp0.expand(B, G))
            prev_image, stride = image[:1], 0
        else:
            stride = S * S * 3
        self._xp_img = d["xp"]
        check(L.dcd_grid_delta_rows(C.byref(self._struct), ptr(image, torch.uint8, B * S * S * 3, "image_u8"),
                                    ptr(prev_image, torch.uint8, S * S * 3 * (B if stride else 1), "prev_u8"), C.c_int64(stride),
                                    ptr(d["cell"], torch.int32), ptr(d["count"], torch.int32), ptr(d["src"], torch.int32),
                                    ptr(d["rows"], torch.float32), ptr(d["tile"], torch.int32), C.c_int32(d["Tt"]),
                                    ptr(d["used"], torch.int32), ptr(d["err"], torch.int32), C.c_int32(B), stream_ptr()))
        if os.environ.get("DCD_TILE_GEMM", "1") != "0":                # fp32 MFMA over the tiles that hold rows, weight blocks read in place
            check(L.dcd_grid_tile_gemm(ptr(d["rows"], torch.float32), ptr(self.wp, torch.float32), ptr(d["tile"], torch.int32),
                                       ptr(d["used"], torch.int32), ptr(d["y"], torch.float32), C.c_int32(d["Tt"]), C.c_int32(F),
                                       C.c_int32(G), stream_ptr()))
        else:                                                          # rocBLAS batched GEMM over ALL tiles on gathered weight blocks
            if "wg" not in d:
                d["wg"] = torch.empty((d["Tt"], F, G), device=self.wp.device)
            torch.index_select(self.wp, 0, d["tile"], out=d["wg"])
            torch.bmm(d["rows"].view(d["Tt"], d["R"], F), d["wg"], out=d["y"].view(d["Tt"], d["R"], G))
        check(L.dcd_gather_sum_rows(ptr(d["y"], torch.float32), ptr(d["src"], torch.int32), ptr(d["xp"], torch.float32), C.c_int32(B),
                                    C.c_int32(9), C.c_int32(G), stream_ptr()))
        return True

    def check_carry_errors(self):
        """Call after a rollout (it synchronises): raises if a step broke the carried projection's one-cell contract."""
        d = getattr(self, "_d", None)
        if d is None:
            return
        e = int(d["err"].item())
        if e:
            d["err"].zero_()
            raise DcdHipError(f"carried teacher projection: error word {e} (1: more than one cell of an observation changed in "
                              "one step, 2: tile table overflow) - the rollout's projections are invalid; DCD_INCREMENTAL_ACT=0 "
                              "multiplies the dense operand instead")

    # ---- the whole step into the storage: carried projection, rest operand, ONE GEMM, cell + heads + draw -------------------
    def begin_rollout(self, num_steps):
        """Once per teacher rollout, OUTSIDE any capture: weights of the last optimiser step, fresh Philox counters."""
        self.refresh()
        if self.fused_heads:
            self.counter_base.add_(int(num_steps))

    @torch.no_grad()
    def act_storage(self, st, step, deterministic=False):
        """`agent.act(obs[step], h[step], masks[step])` + `storage.insert_policy(...)` (adversarial_runner.py:506-517) of a
        teacher step on u8 rollout buffers, nothing read back: writes value_preds[step], actions[step], action_log_probs[step],
        action_log_dist[step], recurrent_hidden_states[step+1].  Needs `fused_heads` and `device_carry_supported`."""
        L, B, H, s = lib(), self.B, self.H, self._struct
        img8 = st.obs_u8["image"]
        self._carry_projection_device(img8[step], img8[step - 1] if step else None)
        scalar = st._obs["time_step"][step] if "time_step" in st._obs else st._obs["direction"][step]
        z = st._obs["random_z"][step] if self.Z else None
        hc_in, masks = st.recurrent_hidden_states[step], st.masks[step]
        check(L.dcd_policy_grid_rest_features(C.byref(s), ptr(scalar, torch.float32, B, "scalar"),
                                              ptr(z, torch.float32, B * self.Z, "random_z") if z is not None else None, C.c_int32(self.Z),
                                              ptr(hc_in, torch.float32, B * 2 * H, "hc_in"), ptr(masks, torch.float32, B, "masks"),
                                              ptr(self.x_rest), C.c_int32(self.KR), C.c_int32(B), stream_ptr()))
        torch.addmm(self._xp_img, self.x_rest, self.w_rest, out=self.gates)
        NA = self._dims["NA"]
        check(L.dcd_policy_wide_cell_heads(C.byref(s), ptr(self.packed), ptr(self.gates), ptr(hc_in), ptr(masks),
                                           ptr(st.recurrent_hidden_states[step + 1], torch.float32, B * 2 * H, "hc_out"),
                                           ptr(st.value_preds[step], torch.float32, B, "value"), ptr(st.actions[step], torch.int64, B, "action"),
                                           ptr(st.action_log_probs[step], torch.float32, B, "log_prob"),
                                           ptr(st.action_log_dist[step], torch.float32, B * NA, "log_dist"), None,
                                           C.c_uint32(self.seed), C.c_uint32(step & 0xFFFFFFFF), C.c_void_p(self.counter_base.data_ptr()),
                                           C.c_int64(self.env_index_base), C.c_int32(int(deterministic)), C.c_int32(B), stream_ptr()))

    def act(self, obs, hc_in, masks, deterministic=False, prev_image=None, first=False):
        """`first` / `prev_image` (the previous step's image of the same episode) enable the carried projection
        (`begin_episode` must have been called); without them every step multiplies the dense operand."""
        if (first or prev_image is not None) and getattr(self, "_incremental", False) and self.device_carry_supported(obs["image"]):
            self._carry_projection_device(obs["image"], None if first else prev_image)
            return self._act_carried(obs, hc_in, masks, deterministic)
        if (first or prev_image is not None) and getattr(self, "_incremental", False) and not torch.cuda.is_current_stream_capturing():
            self._incremental = self._carry_projection(obs["image"], None if first else prev_image)
            if self._incremental:
                return self._act_carried(obs, hc_in, masks, deterministic)
        self._incremental = False                            # a step that was not carried breaks the chain for this episode
        return self._act_dense(obs, hc_in, masks, deterministic)

    def _act_carried(self, obs, hc_in, masks, deterministic):
        from models.common import one_hot
        net, H = self.net, self.H
        scalar = obs.get("direction")
        if scalar is None:
            scalar = obs.get("time_step")
        rest = [net.scalar_embed(one_hot(net.scalar_dim, scalar))]
        if self.Z:
            rest.append(obs["random_z"])
        rest.append(hc_in[:, :H] * masks)
        FP = self.K - self.H - self.Z - self._dims["E"]
        self.gates = torch.addmm(self._xp_img, torch.cat(rest, dim=1), self.w_t[FP:self.K]) + self.bias
        return self._finish(hc_in, masks, deterministic)

    def _finish(self, hc_in, masks, deterministic):
        H = self.H
        i, f, g, o = self.gates.split(H, dim=1)
        c = torch.sigmoid(f) * (hc_in[:, H:] * masks) + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        dist = self.net.actor(h)
        value = self.net.critic(h)
        action = dist.mode() if deterministic else dist.sample()
        return value, action, dist.logits, (h, c)

    def _act_dense(self, obs, hc_in, masks, deterministic=False):
        """`net.act(obs, (h, c), masks)` (multigrid_models.py:141-167): value [B,1], action [B,1], normalised logits
        [B,A], (h', c').  obs: dict with image / time_step (or direction) / random_z; hc_in: [B, 2H] = h | c."""
        B, H, s = self.B, self.H, self._struct
        scalar = obs.get("direction")
        if scalar is None:
            scalar = obs.get("time_step")
        z = obs.get("random_z") if self.Z else None
        u8 = obs["image"].dtype == torch.uint8
        fn = lib().dcd_policy_grid_features_u8 if u8 else lib().dcd_policy_grid_features
        check(fn(
            C.byref(s), ptr(obs["image"], torch.uint8 if u8 else torch.float32, B * 3 * self._dims["V"] ** 2, "image"), ptr(scalar, torch.float32, B, "scalar"),
            ptr(z, torch.float32, B * self.Z, "random_z") if z is not None else None, C.c_int32(self.Z),
            ptr(hc_in, torch.float32, B * 2 * H, "hc_in"), ptr(masks, torch.float32, B, "masks"), ptr(self.x), C.c_int32(self.KP),
            C.c_int32(B), stream_ptr()))
        if self.gates.data_ptr() == 0 or self.gates.shape != (B, 4 * H):
            self.gates = torch.empty((B, 4 * H), device=self.x.device)
        torch.addmm(self.bias, self.x, self.w_t, out=self.gates)
        return self._finish(hc_in, masks, deterministic)
