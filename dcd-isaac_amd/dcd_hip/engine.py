"""MultiGridEngine: one batch of B MultiGrid envs resident in HBM, driven through the C ABI.

Thin host object around a `DcdMg*` handle (include/dcd_hip.h).  All arguments and results are
CUDA(=HIP) tensors; calls are asynchronous on torch's current stream and never synchronise,
except `poll_errors()` / `raise_errors()`.
"""
import ctypes as C

import torch

from . import (AdvObs, DcdHipError, ERR_BITS, MgConfig, StepOut, StudentObs, check, lib, ptr, stream_ptr)

_CFG_DEFAULTS = dict(size=15, agent_view_size=5, max_steps=250, max_episode_steps=250, n_clutter=50,
                     resample_n_clutter=0, choose_goal_last=0, see_through_walls=1, random_z_dim=50,
                     editor_actions=2, fixed_environment=0, goal_noise=0.0, engine_seed=0, env_index_base=0, replicas=1)

_STUDENT_KEYS = ("image_u8", "dir_u8", "image_f32", "dir_f32")
_ADV_KEYS = ("image_u8", "image_f32", "time_step", "random_z")
_STEP_KEYS = ("reward", "done", "info", "ep_ret", "ep_len", "masks", "bad_masks", "cliffhanger_masks")


class MultiGridEngine:
    def __init__(self, num_envs, device="cuda:0", **cfg):
        if not torch.cuda.is_available():
            raise DcdHipError("MultiGridEngine needs a HIP device (torch.cuda.is_available() is False); "
                              "there is no CPU path")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise DcdHipError(f"MultiGridEngine: device must be a cuda/hip device, got {device}")
        full = dict(_CFG_DEFAULTS)
        unknown = set(cfg) - set(full)
        if unknown:
            raise TypeError(f"unknown MultiGridEngine config keys: {sorted(unknown)}")
        full.update(cfg)
        self.cfg = MgConfig(num_envs=int(num_envs), **{k: (float(v) if k == "goal_noise" else int(v))
                                                        for k, v in full.items()})
        self.B, self.S, self.V, self.Z = int(num_envs), self.cfg.size, self.cfg.agent_view_size, self.cfg.random_z_dim
        self.n_adv_actions = (self.S - 2) ** 2
        h = C.c_void_p()
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        check(lib().dcd_mg_create(C.byref(self.cfg), C.c_int(index), C.byref(h)))
        self._h = h
        self._views = []

    def replica(self, k):
        """Replica k (1 <= k < replicas) of this batch as an engine of its own: a second copy of the B env records in
        the same allocation (dcd_mg_replica).  `clone_levels` copies a level set into it, `step_pair` advances two
        consecutive replicas in one launch."""
        h = C.c_void_p()
        check(lib().dcd_mg_replica(self._h, C.c_int32(int(k)), C.byref(h)))
        v = object.__new__(MultiGridEngine)
        v.device, v.cfg, v.B, v.S, v.V, v.Z, v.n_adv_actions = self.device, self.cfg, self.B, self.S, self.V, self.Z, self.n_adv_actions
        v._h, v._views, v._owner = h, [], self
        self._views.append(v)
        return v

    def clone_levels(self, src):
        """self := src (grids, agent / episode state, Philox keys and counters, metrics)."""
        check(lib().dcd_mg_clone_levels(self._h, src._h, stream_ptr()))

    def step_pair(self, other, actions, other_actions, flags=0, out=None, other_out=None):
        """dcd_mg_step of this replica and the next one (`other`) in ONE launch."""
        st0, st1 = self._step_struct(out), self._step_struct(other_out)
        check(lib().dcd_mg_step_pair(self._h, other._h, ptr(actions, torch.int64, self.B, "actions"),
                                     ptr(other_actions, torch.int64, self.B, "other_actions"), C.c_int32(flags),
                                     C.byref(st0), C.byref(st1), stream_ptr()))
        return out, other_out

    def close(self):
        for v in getattr(self, "_views", []):
            v.close()
        if getattr(self, "_h", None):
            lib().dcd_mg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- buffers -----------------------------------------------------------------------
    def student_buffers(self, n=None, u8=True, f32=True):
        n = self.B if n is None else n
        V, d = self.V, self.device
        out = {}
        if u8:
            out["image_u8"] = torch.zeros((n, V, V, 3), dtype=torch.uint8, device=d)
            out["dir_u8"] = torch.zeros((n,), dtype=torch.uint8, device=d)
        if f32:
            out["image_f32"] = torch.zeros((n, 3, V, V), dtype=torch.float32, device=d)
            out["dir_f32"] = torch.zeros((n, 1), dtype=torch.float32, device=d)
        return out

    def adv_buffers(self, u8=True, f32=True):
        B, S, d = self.B, self.S, self.device
        out = {"time_step": torch.zeros((B, 1), dtype=torch.float32, device=d),
               "random_z": torch.zeros((B, self.Z), dtype=torch.float32, device=d),
               "done": torch.zeros((B,), dtype=torch.uint8, device=d)}
        if u8:
            out["image_u8"] = torch.zeros((B, S, S, 3), dtype=torch.uint8, device=d)
        if f32:
            out["image_f32"] = torch.zeros((B, 3, S, S), dtype=torch.float32, device=d)
        return out

    def step_buffers(self, u8=True, f32=True):
        B, d = self.B, self.device
        return dict(obs=self.student_buffers(u8=u8, f32=f32), term_obs=self.student_buffers(u8=u8, f32=f32),
                    reward=torch.zeros((B,), dtype=torch.float32, device=d),
                    done=torch.zeros((B,), dtype=torch.uint8, device=d),
                    info=torch.zeros((B,), dtype=torch.uint8, device=d),
                    ep_ret=torch.zeros((B,), dtype=torch.float32, device=d),
                    ep_len=torch.zeros((B,), dtype=torch.int32, device=d),
                    masks=torch.zeros((B,), dtype=torch.float32, device=d),
                    bad_masks=torch.zeros((B,), dtype=torch.float32, device=d),
                    cliffhanger_masks=torch.zeros((B,), dtype=torch.float32, device=d))

    def _student_struct(self, bufs, n):
        if bufs is None:
            return StudentObs()
        V = self.V
        return StudentObs(ptr(bufs.get("image_u8"), torch.uint8, n * V * V * 3, "image_u8"),
                          ptr(bufs.get("dir_u8"), torch.uint8, n, "dir_u8"),
                          ptr(bufs.get("image_f32"), torch.float32, n * 3 * V * V, "image_f32"),
                          ptr(bufs.get("dir_f32"), torch.float32, n, "dir_f32"))

    def _adv_struct(self, bufs):
        if bufs is None:
            return AdvObs()
        B, S = self.B, self.S
        return AdvObs(ptr(bufs.get("image_u8"), torch.uint8, B * S * S * 3, "image_u8"),
                      ptr(bufs.get("image_f32"), torch.float32, B * 3 * S * S, "image_f32"),
                      ptr(bufs.get("time_step"), torch.float32, B, "time_step"),
                      ptr(bufs.get("random_z"), torch.float32, B * self.Z, "random_z"))

    def _step_struct(self, out):
        B = self.B
        dt = dict(reward=torch.float32, done=torch.uint8, info=torch.uint8, ep_ret=torch.float32, ep_len=torch.int32,
                  masks=torch.float32, bad_masks=torch.float32, cliffhanger_masks=torch.float32)
        p = {k: ptr(out.get(k), dt[k], B, k) for k in _STEP_KEYS}
        return StepOut(self._student_struct(out.get("obs"), B), p["reward"], p["done"], p["info"], p["ep_ret"],
                       p["ep_len"], self._student_struct(out.get("term_obs"), B), p["masks"], p["bad_masks"],
                       p["cliffhanger_masks"])

    # ---- errors ------------------------------------------------------------------------
    def poll_errors(self):
        bits, first = C.c_uint32(0), C.c_int32(-1)
        check(lib().dcd_mg_poll_errors(self._h, C.byref(bits), C.byref(first), stream_ptr()))
        return int(bits.value), int(first.value)

    def raise_errors(self):
        """Synchronises; raises the reference's exception type for latched device errors."""
        bits, first = self.poll_errors()
        if bits:
            msgs = [m for b, m in ERR_BITS.items() if bits & b]
            exc = AssertionError if bits & (1 | 32) and not bits & ~(1 | 32) else ValueError
            raise exc(f"{'; '.join(msgs)} (first env {first})")

    # ---- calls -------------------------------------------------------------------------
    def set_seed(self, seeds):
        if not isinstance(seeds, torch.Tensor):
            seeds = torch.as_tensor(list(seeds), dtype=torch.int64)
        seeds = seeds.to(device=self.device, dtype=torch.int64).contiguous()
        check(lib().dcd_mg_set_seed(self._h, ptr(seeds, torch.int64, self.B, "seeds"), stream_ptr()))

    def reset(self, out=None):
        out = self.adv_buffers() if out is None else out
        st = self._adv_struct(out)
        check(lib().dcd_mg_reset(self._h, C.byref(st), stream_ptr()))
        return out

    def step_adversary(self, loc, out=None):
        out = self.adv_buffers() if out is None else out
        st = self._adv_struct(out)
        check(lib().dcd_mg_step_adversary(self._h, ptr(loc, torch.int64, self.B, "loc"), C.byref(st),
                                          ptr(out.get("done"), torch.uint8, self.B, "done"), stream_ptr()))
        return out

    def reset_agent(self, out=None):
        out = self.student_buffers() if out is None else out
        st = self._student_struct(out, self.B)
        check(lib().dcd_mg_reset_agent(self._h, C.byref(st), stream_ptr()))
        return out

    def reset_random(self, out=None):
        out = self.student_buffers() if out is None else out
        st = self._student_struct(out, self.B)
        check(lib().dcd_mg_reset_random(self._h, C.byref(st), stream_ptr()))
        return out

    def reset_to_encoding(self, enc, idx=None, out=None):
        n = enc.shape[0]
        out = self.student_buffers(n) if out is None else out
        st = self._student_struct(out, n)
        check(lib().dcd_mg_reset_to_encoding(
            self._h, ptr(enc, torch.uint8, n * self.S * self.S * 3, "enc"),
            ptr(idx, torch.int32, n, "idx"), C.c_int32(n), C.byref(st), stream_ptr()))
        return out

    def reset_to_fixed_level(self, enc, idx=None, out=None):
        n = enc.shape[0]
        out = self.student_buffers(n) if out is None else out
        st = self._student_struct(out, n)
        check(lib().dcd_mg_reset_to_fixed_level(
            self._h, ptr(enc, torch.uint8, n * self.S * self.S * 3, "enc"),
            ptr(idx, torch.int32, n, "idx"), C.c_int32(n), C.byref(st), stream_ptr()))
        return out

    def reset_to_actions(self, actions, lens, idx=None, out=None):
        n, max_len = actions.shape
        out = self.student_buffers(n) if out is None else out
        st = self._student_struct(out, n)
        check(lib().dcd_mg_reset_to_actions(
            self._h, ptr(actions, torch.int64, n * max_len, "actions"), ptr(lens, torch.int32, n, "lens"),
            C.c_int32(max_len), ptr(idx, torch.int32, n, "idx"), C.c_int32(n), C.byref(st), stream_ptr()))
        return out

    def reset_to_table_rows(self, table, table_lens, level_idx, out):
        """Masked reset from a device level table (dcd_mg_reset_to_encoding_rows / _actions_rows): env i takes level
        table[level_idx[i]] where level_idx[i] >= 0, everything else is left alone; `out` rows are indexed by env."""
        st = self._student_struct(out, self.B)
        if table.dtype == torch.uint8:
            check(lib().dcd_mg_reset_to_encoding_rows(self._h, ptr(table, torch.uint8, name="table"), ptr(level_idx, torch.int32, self.B, "level_idx"),
                                                      C.byref(st), stream_ptr()))
        else:
            check(lib().dcd_mg_reset_to_actions_rows(self._h, ptr(table, torch.int64, name="table"), ptr(table_lens, torch.int32, table.shape[0], "table_lens"),
                                                     C.c_int32(table.shape[1]), ptr(level_idx, torch.int32, self.B, "level_idx"), C.byref(st), stream_ptr()))
        return out

    def mutate_level(self, num_edits, out=None):
        out = self.student_buffers() if out is None else out
        st = self._student_struct(out, self.B)
        check(lib().dcd_mg_mutate_level(self._h, C.c_int32(int(num_edits)), C.byref(st), stream_ptr()))
        return out

    def step(self, actions, flags=0, out=None):
        out = self.step_buffers() if out is None else out
        st = self._step_struct(out)
        check(lib().dcd_mg_step(self._h, ptr(actions, torch.int64, self.B, "actions"), C.c_int32(flags),
                                C.byref(st), stream_ptr()))
        return out

    def step_random(self, action_seed, t0, K, flags=0, out=None, n_episodes=None):
        st = self._step_struct(out) if out is not None else StepOut()
        check(lib().dcd_mg_step_random(self._h, C.c_uint32(action_seed), C.c_uint32(t0), C.c_int32(K),
                                       C.c_int32(flags), C.byref(st),
                                       ptr(n_episodes, torch.int64, 1, "n_episodes"), stream_ptr()))
        return out

    @staticmethod
    def step_epb(rows):
        """Envs per workgroup the step kernel picks for a launch over `rows` env records (mg_kernels.hip::step_epb)."""
        return 64 if rows >= 64 * 512 else (32 if rows >= 32 * 512 else 16)

    def profile_begin(self, capacity):
        check(lib().dcd_mg_profile_begin(self._h, C.c_int32(capacity)))

    def profile_end(self):
        """Synchronises; returns (launches, total kernel milliseconds) of the bracketed dcd_mg_step calls."""
        n, ms = C.c_int32(0), C.c_double(0.0)
        check(lib().dcd_mg_profile_end(self._h, C.byref(n), C.byref(ms), stream_ptr()))
        return int(n.value), float(ms.value)

    def get_encodings(self):
        out = torch.empty((self.B, self.S, self.S, 3), dtype=torch.uint8, device=self.device)
        check(lib().dcd_mg_get_encodings(self._h, ptr(out), stream_ptr()))
        return out

    def get_metrics(self):
        m = [torch.empty((self.B,), dtype=torch.int32, device=self.device) for _ in range(4)]
        check(lib().dcd_mg_get_metrics(self._h, *[ptr(t) for t in m], stream_ptr()))
        return dict(n_clutter_placed=m[0], passable=m[1], shortest_path_length=m[2], distance_to_goal=m[3])

    def get_state(self):
        out = torch.empty((self.B, 16), dtype=torch.int32, device=self.device)
        check(lib().dcd_mg_get_state(self._h, ptr(out), stream_ptr()))
        return out


def random_actions(B, env_index_base, seed, t, num_actions, device, out=None, T=1):
    """Synthetic Philox actions: int64[B] (T == 1) or int64[T,B] for steps t .. t+T-1."""
    if out is None:
        out = torch.empty((B,) if T == 1 else (T, B), dtype=torch.int64, device=device)
    check(lib().dcd_random_actions(ptr(out, torch.int64, B * T, "actions"), C.c_int32(B), C.c_int32(T),
                                   C.c_int64(env_index_base), C.c_uint32(seed), C.c_uint32(t), C.c_int32(num_actions),
                                   stream_ptr()))
    return out
