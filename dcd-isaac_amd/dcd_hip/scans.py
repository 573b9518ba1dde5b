"""Tensor-level wrappers of the scan / scoring entry points of the C ABI (include/dcd_hip.h).

Inputs are the [T(+1), B(, 1)] CUDA tensors of a RolloutStorage (time-major, env axis
contiguous); trailing singleton dims are accepted.  Nothing here synchronises except
`episode_records`, which must read the record count back.
"""
import ctypes as C

import numpy as np
import torch

from . import DcdHipError, EPISODE_REC_WORDS, check, lib, ptr, stream_ptr

EPISODE_REC = np.dtype([("actor", "<i4"), ("seed", "<i4"), ("start_t", "<i4"), ("num_steps", "<i4"),
                        ("mean_score", "<f4"), ("max_score", "<f4"), ("reward_sum", "<f4"), ("flags", "<i4")])


def _tb(t, rows, B, name, dtype=torch.float32):
    if t is None:
        return None
    if t.dim() == 3 and t.shape[-1] == 1:
        t = t.squeeze(-1)
    if t.dim() != 2 or t.shape[0] != rows or t.shape[1] != B:
        raise DcdHipError(f"{name}: expected shape [{rows},{B}(,1)], got {tuple(t.shape)}")
    return ptr(t, dtype, rows * B, name)


def gae_returns(rewards, value_preds, masks, returns, gamma, gae_lambda):
    """algos/storage.py:233-256 — writes returns[0:T]."""
    T, B = rewards.shape[0], rewards.shape[1]
    check(lib().dcd_gae_returns(_tb(rewards, T, B, "rewards"), _tb(value_preds, T + 1, B, "value_preds"),
                                _tb(masks, T + 1, B, "masks"), _tb(returns, T + 1, B, "returns"),
                                C.c_int32(T), C.c_int32(B), C.c_double(gamma), C.c_double(gae_lambda), stream_ptr()))
    return returns


def _ptr_array(ptrs):
    return (C.c_void_p * len(ptrs))(*[p.value if p is not None else None for p in ptrs])


def gae_returns_batch(sets, gamma, gae_lambda):
    """`gae_returns` for up to two rollouts of one shape in ONE launch (PAIRED's protagonist and antagonist):
    sets = [(rewards, value_preds, masks, returns), ...]."""
    T, B = sets[0][0].shape[0], sets[0][0].shape[1]
    cols = [[_tb(st[0], T, B, "rewards") for st in sets], [_tb(st[1], T + 1, B, "value_preds") for st in sets],
            [_tb(st[2], T + 1, B, "masks") for st in sets], [_tb(st[3], T + 1, B, "returns") for st in sets]]
    check(lib().dcd_gae_returns_batch(C.c_int32(len(sets)), *[_ptr_array(c) for c in cols], C.c_int32(T), C.c_int32(B),
                                      C.c_double(gamma), C.c_double(gae_lambda), stream_ptr()))


def discounted_returns(rewards, value_preds, masks, returns, gamma):
    """algos/storage.py:258-276 — writes returns[0:T+1]."""
    T, B = rewards.shape[0], rewards.shape[1]
    check(lib().dcd_discounted_returns(_tb(rewards, T, B, "rewards"), _tb(value_preds, T + 1, B, "value_preds"),
                                       _tb(masks, T + 1, B, "masks"), _tb(returns, T + 1, B, "returns"),
                                       C.c_int32(T), C.c_int32(B), C.c_double(gamma), stream_ptr()))
    return returns


def batched_value_loss(returns, value_preds, mode=0, power=1, clipped=True, out=None):
    """algos/storage.py:290-327 — mode 0 abs / 1 signed / 2 positive-only; returns [B,1]."""
    T, B = returns.shape[0] - 1, returns.shape[1]
    out = torch.empty((B, 1), dtype=torch.float32, device=returns.device) if out is None else out
    check(lib().dcd_batched_value_loss(_tb(returns, T + 1, B, "returns"), _tb(value_preds, T + 1, B, "value_preds"),
                                       ptr(out, torch.float32, B, "out"), C.c_int32(T), C.c_int32(B), C.c_int32(mode),
                                       C.c_int32(power), C.c_int32(int(clipped)), stream_ptr()))
    return out


class ScoreArgs(C.Structure):          # DcdEpisodeScoreArgs
    _fields_ = [(k, C.c_void_p) for k in ("returns", "value_preds", "rewards", "masks", "cliffhanger_masks", "level_seeds",
                                          "action_log_dist", "recs")] + [("max_recs", C.c_int32), ("n_recs", C.c_void_p), ("workspace", C.c_void_p)]


class EpisodeScorer:
    """Owns the record / workspace buffers for dcd_episode_scores on one device."""

    def __init__(self, T, B, device):
        self.T, self.B, self.device = T, B, device
        self.max_recs = B * (T + 1)
        self.recs = torch.zeros((self.max_recs, EPISODE_REC_WORDS), dtype=torch.int32, device=device)
        self.n_recs = torch.zeros((1,), dtype=torch.int32, device=device)
        self.workspace = torch.zeros((2 * B + 64,), dtype=torch.int32, device=device)

    def launch(self, returns, value_preds, rewards, masks, cliffhanger_masks, level_seeds, action_log_dist,
               strategy, gamma):
        T, B = self.T, self.B
        A = 0
        ld = None
        if action_log_dist is not None:
            A = action_log_dist.shape[-1]
            ld = ptr(action_log_dist, torch.float32, T * B * A, "action_log_dist")
        check(lib().dcd_episode_scores(
            _tb(returns, T + 1, B, "returns"), _tb(value_preds, T + 1, B, "value_preds"),
            _tb(rewards, T, B, "rewards"), _tb(masks, T + 1, B, "masks"),
            _tb(cliffhanger_masks, T + 1, B, "cliffhanger_masks"), _tb(level_seeds, T, B, "level_seeds", torch.int32),
            ld, C.c_int32(T), C.c_int32(B), C.c_int32(A), C.c_int32(strategy), C.c_double(gamma),
            ptr(self.recs), C.c_int32(self.max_recs), ptr(self.n_recs), ptr(self.workspace), stream_ptr()))

    @staticmethod
    def launch_batch(scorers, argsets, strategy, gamma):
        """`launch` of up to two scorers of one (T, B) on their own rollouts in one launch sequence:
        argsets[k] = (returns, value_preds, rewards, masks, cliffhanger_masks, level_seeds, action_log_dist)."""
        T, B = scorers[0].T, scorers[0].B
        arr = (ScoreArgs * len(scorers))()
        A = 0
        for k, (sc, a) in enumerate(zip(scorers, argsets)):
            ld = None
            if a[6] is not None:
                A = a[6].shape[-1]
                ld = ptr(a[6], torch.float32, T * B * A, "action_log_dist")
            vals = [_tb(a[0], T + 1, B, "returns"), _tb(a[1], T + 1, B, "value_preds"), _tb(a[2], T, B, "rewards"),
                    _tb(a[3], T + 1, B, "masks"), _tb(a[4], T + 1, B, "cliffhanger_masks"), _tb(a[5], T, B, "level_seeds", torch.int32), ld,
                    ptr(sc.recs)]
            for name, v in zip(("returns", "value_preds", "rewards", "masks", "cliffhanger_masks", "level_seeds", "action_log_dist", "recs"), vals):
                setattr(arr[k], name, v.value if v is not None else None)
            arr[k].max_recs, arr[k].n_recs, arr[k].workspace = sc.max_recs, ptr(sc.n_recs).value, ptr(sc.workspace).value
        check(lib().dcd_episode_scores_batch(C.c_int32(len(scorers)), arr, C.c_int32(T), C.c_int32(B), C.c_int32(A), C.c_int32(strategy),
                                             C.c_double(gamma), stream_ptr()))

    @staticmethod
    def _arg_array(scorers, argsets):
        T, B = scorers[0].T, scorers[0].B
        arr = (ScoreArgs * len(scorers))()
        A = 0
        for k, (sc, a) in enumerate(zip(scorers, argsets)):
            ld = None
            if a[6] is not None:
                A = a[6].shape[-1]
                ld = ptr(a[6], torch.float32, T * B * A, "action_log_dist")
            vals = [_tb(a[0], T + 1, B, "returns"), _tb(a[1], T + 1, B, "value_preds"), _tb(a[2], T, B, "rewards"),
                    _tb(a[3], T + 1, B, "masks"), _tb(a[4], T + 1, B, "cliffhanger_masks"), _tb(a[5], T, B, "level_seeds", torch.int32), ld,
                    ptr(sc.recs)]
            for name, v in zip(("returns", "value_preds", "rewards", "masks", "cliffhanger_masks", "level_seeds", "action_log_dist", "recs"), vals):
                setattr(arr[k], name, v.value if v is not None else None)
            arr[k].max_recs, arr[k].n_recs, arr[k].workspace = sc.max_recs, ptr(sc.n_recs).value, ptr(sc.workspace).value
        return arr, A

    def records(self):
        """Synchronises: returns the records as a numpy structured array (reference order)."""
        n = int(self.n_recs.item())
        return self.recs[:n].cpu().numpy().view(EPISODE_REC).reshape(n)

    def grounded(self, n, grounded_values, value_preds, positive_only, dense):
        """Second pass for grounded_* strategies; grounded_values float32[n] on device."""
        mean = torch.empty((n,), dtype=torch.float32, device=self.device)
        mx = torch.empty((n,), dtype=torch.float32, device=self.device)
        if n:
            check(lib().dcd_grounded_scores(ptr(self.recs), C.c_int32(n), ptr(grounded_values, torch.float32, n, "grounded"),
                                            _tb(value_preds, self.T + 1, self.B, "value_preds"), C.c_int32(self.B),
                                            C.c_int32(int(positive_only)), C.c_int32(int(dense)), ptr(mean), ptr(mx),
                                            stream_ptr()))
        return mean, mx


def rollout_return_stats(ep_ret, info, mean_out=None, max_out=None, n_out=None):
    """envs/runners/adversarial_runner.py:257-271 — ep_ret [T,B] fp32, info [T,B] uint8."""
    T, B = ep_ret.shape
    d = ep_ret.device
    mean_out = torch.empty((B, 1), dtype=torch.float32, device=d) if mean_out is None else mean_out
    max_out = torch.empty((B, 1), dtype=torch.float32, device=d) if max_out is None else max_out
    n_out = torch.empty((B,), dtype=torch.int32, device=d) if n_out is None else n_out
    check(lib().dcd_rollout_return_stats(ptr(ep_ret, torch.float32, T * B, "ep_ret"), ptr(info, torch.uint8, T * B, "info"),
                                         C.c_int32(T), C.c_int32(B), ptr(mean_out, torch.float32, B), ptr(max_out, torch.float32, B),
                                         ptr(n_out, torch.int32, B), stream_ptr()))
    return mean_out, max_out, n_out


def rollout_return_stats_batch(sets):
    """`rollout_return_stats` for up to two rollouts in one launch: sets = [(ep_ret, info, mean_out, max_out, n_out), ...]."""
    T, B = sets[0][0].shape
    cols = [[ptr(st[0], torch.float32, T * B, "ep_ret") for st in sets], [ptr(st[1], torch.uint8, T * B, "info") for st in sets],
            [ptr(st[2], torch.float32, B) for st in sets], [ptr(st[3], torch.float32, B) for st in sets], [ptr(st[4], torch.int32, B) for st in sets]]
    a = [_ptr_array(c) for c in cols]
    check(lib().dcd_rollout_return_stats_batch(C.c_int32(len(sets)), a[0], a[1], C.c_int32(T), C.c_int32(B), a[2], a[3], a[4], stream_ptr()))


def env_return(agent_mean, agent_max, adv_mean, adv_max, mode=0, out=None):
    """envs/runners/adversarial_runner.py:637-664 — mode 0 paired, 1 flexible_paired, 2 minimax."""
    B = agent_max.numel()
    out = torch.empty_like(agent_max) if out is None else out
    check(lib().dcd_env_return(ptr(agent_mean, torch.float32, B), ptr(agent_max, torch.float32, B),
                               ptr(adv_mean, torch.float32, B), ptr(adv_max, torch.float32, B),
                               ptr(out, torch.float32, B), C.c_int32(B), C.c_int32(mode), stream_ptr()))
    return out


def rollout_tail_batch(gae_sets, stat_sets, gamma, gae_lambda, scorers=None, score_argsets=None, strategy=0):
    """dcd_rollout_tail_batch: what follows a pair of rollouts in two launches (one without scoring) —
    gae_sets = [(rewards, value_preds, masks, returns), ...] as `gae_returns_batch`, stat_sets = [(ep_ret, info, mean_out, max_out,
    n_out), ...] as `rollout_return_stats_batch`, and optionally the scorers / argument sets of `EpisodeScorer.launch_batch`."""
    T, B = gae_sets[0][0].shape[0], gae_sets[0][0].shape[1]
    n = len(gae_sets)
    g = [[_tb(st[0], T, B, "rewards") for st in gae_sets], [_tb(st[1], T + 1, B, "value_preds") for st in gae_sets],
         [_tb(st[2], T + 1, B, "masks") for st in gae_sets], [_tb(st[3], T + 1, B, "returns") for st in gae_sets]]
    q = [[ptr(st[0], torch.float32, T * B, "ep_ret") for st in stat_sets], [ptr(st[1], torch.uint8, T * B, "info") for st in stat_sets],
         [ptr(st[2], torch.float32, B) for st in stat_sets], [ptr(st[3], torch.float32, B) for st in stat_sets],
         [ptr(st[4], torch.int32, B) for st in stat_sets]]
    ga, qa = [_ptr_array(c) for c in g], [_ptr_array(c) for c in q]
    arr, A = (None, 0)
    if scorers is not None:
        arr, A = EpisodeScorer._arg_array(scorers, score_argsets)
    check(lib().dcd_rollout_tail_batch(C.c_int32(n), ga[0], ga[1], ga[2], ga[3], qa[0], qa[1], qa[2], qa[3], qa[4], arr, C.c_int32(T), C.c_int32(B),
                                       C.c_int32(A), C.c_int32(strategy), C.c_double(gamma), C.c_double(gae_lambda), stream_ptr()))
