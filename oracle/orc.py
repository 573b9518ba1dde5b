"""ctypes binding of the CPU oracle (oracle/liborc.so) on numpy arrays.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never import this from dcd-isaac_amd/.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liborc.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("mg_oracle.c", "scan_oracle.c")]
    hdr = os.path.join(_HERE, "..", "include", "dcd_hip.h")
    if not force and os.path.exists(_LIB) and all(
            os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in src + [hdr] if os.path.exists(s)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return _LIB


class MgConfig(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("size", C.c_int32), ("agent_view_size", C.c_int32),
                ("max_steps", C.c_int32), ("max_episode_steps", C.c_int32), ("n_clutter", C.c_int32),
                ("resample_n_clutter", C.c_int32), ("choose_goal_last", C.c_int32),
                ("see_through_walls", C.c_int32), ("random_z_dim", C.c_int32),
                ("editor_actions", C.c_int32), ("fixed_environment", C.c_int32),
                ("goal_noise", C.c_float), ("engine_seed", C.c_uint32), ("env_index_base", C.c_int64),
                ("replicas", C.c_int32), ("reserved", C.c_int32)]


class StudentObs(C.Structure):
    _fields_ = [("image_u8", C.c_void_p), ("dir_u8", C.c_void_p),
                ("image_f32", C.c_void_p), ("dir_f32", C.c_void_p)]


class AdvObs(C.Structure):
    _fields_ = [("image_u8", C.c_void_p), ("image_f32", C.c_void_p),
                ("time_step", C.c_void_p), ("random_z", C.c_void_p)]


class StepOut(C.Structure):
    _fields_ = [("obs", StudentObs), ("reward", C.c_void_p), ("done", C.c_void_p), ("info", C.c_void_p),
                ("ep_ret", C.c_void_p), ("ep_len", C.c_void_p), ("term_obs", StudentObs),
                ("masks", C.c_void_p), ("bad_masks", C.c_void_p), ("cliffhanger_masks", C.c_void_p)]


EPISODE_REC = np.dtype([("actor", "<i4"), ("seed", "<i4"), ("start_t", "<i4"), ("num_steps", "<i4"),
                        ("mean_score", "<f4"), ("max_score", "<f4"), ("reward_sum", "<f4"),
                        ("flags", "<i4")])

DEFAULT_CFG = dict(num_envs=1, size=15, agent_view_size=5, max_steps=250, max_episode_steps=250,
                   n_clutter=25, resample_n_clutter=0, choose_goal_last=1, see_through_walls=1,
                   random_z_dim=50, editor_actions=2, fixed_environment=0, goal_noise=0.0,
                   engine_seed=0, env_index_base=0)


def make_config(**kw):
    d = dict(DEFAULT_CFG)
    d.update(kw)
    return MgConfig(**d)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_mg_create.restype = C.c_void_p
        _lib.orc_mg_poll_errors.restype = C.c_uint32
        _lib.orc_episode_scores.restype = C.c_int32
        _lib.orc_mg_rollout_range.restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class OracleMg:
    """Batch of scalar oracle envs with the same call set as the HIP engine."""

    _prefix = "orc_mg_"

    @staticmethod
    def _library():
        return lib()

    def _fn(self, name):
        return getattr(self._library(), self._prefix + name)

    def __init__(self, **cfg):
        self.cfg = make_config(**cfg)
        self.B = self.cfg.num_envs
        self.S = self.cfg.size
        self.V = self.cfg.agent_view_size
        self.Z = self.cfg.random_z_dim
        self.h = C.c_void_p(self._fn("create")(C.byref(self.cfg)))
        if not self.h:
            raise ValueError("orc_mg_create rejected the config")

    def __del__(self):
        if getattr(self, "h", None):
            self._fn("destroy")(self.h)
            self.h = None

    def clone_levels(self, src):
        """self := src, every env object with all of its state (the second copy of a PAIRED level set)."""
        self._fn("clone_levels")(self.h, src.h)

    # -- buffers
    def _student(self, n=None):
        n = self.B if n is None else n
        V = self.V
        bufs = dict(image_u8=np.zeros((n, V, V, 3), np.uint8), dir_u8=np.zeros((n,), np.uint8),
                    image_f32=np.zeros((n, 3, V, V), np.float32), dir_f32=np.zeros((n, 1), np.float32))
        st = StudentObs(*[_p(bufs[k]) for k in ("image_u8", "dir_u8", "image_f32", "dir_f32")])
        return bufs, st

    def _adv(self):
        B, S = self.B, self.S
        bufs = dict(image_u8=np.zeros((B, S, S, 3), np.uint8), image_f32=np.zeros((B, 3, S, S), np.float32),
                    time_step=np.zeros((B, 1), np.float32), random_z=np.zeros((B, self.Z), np.float32))
        st = AdvObs(*[_p(bufs[k]) for k in ("image_u8", "image_f32", "time_step", "random_z")])
        return bufs, st

    # -- calls
    def set_seed(self, seeds):
        s = np.ascontiguousarray(seeds, np.int64)
        self._fn("set_seed")(self.h, _p(s))

    def poll_errors(self):
        first = C.c_int32(-1)
        bits = self._fn("poll_errors")(self.h, C.byref(first))
        return int(bits), int(first.value)

    def reset(self):
        bufs, st = self._adv()
        self._fn("reset")(self.h, C.byref(st))
        return bufs

    def step_adversary(self, loc):
        loc = np.ascontiguousarray(loc, np.int64)
        bufs, st = self._adv()
        done = np.zeros((self.B,), np.uint8)
        self._fn("step_adversary")(self.h, _p(loc), C.byref(st), _p(done))
        bufs["done"] = done
        return bufs

    def reset_agent(self):
        bufs, st = self._student()
        self._fn("reset_agent")(self.h, C.byref(st))
        return bufs

    def reset_random(self):
        bufs, st = self._student()
        self._fn("reset_random")(self.h, C.byref(st))
        return bufs

    def reset_to_encoding(self, enc, idx=None):
        enc = np.ascontiguousarray(enc, np.uint8)
        n = enc.shape[0]
        idx_a = None if idx is None else np.ascontiguousarray(idx, np.int32)
        bufs, st = self._student(n)
        self._fn("reset_to_encoding")(self.h, _p(enc), _p(idx_a), C.c_int32(n), C.byref(st))
        return bufs

    def reset_to_fixed_level(self, enc, idx=None):
        enc = np.ascontiguousarray(enc, np.uint8)
        n = enc.shape[0]
        idx_a = None if idx is None else np.ascontiguousarray(idx, np.int32)
        bufs, st = self._student(n)
        self._fn("reset_to_fixed_level")(self.h, _p(enc), _p(idx_a), C.c_int32(n), C.byref(st))
        return bufs

    def reset_to_actions(self, actions, lens, idx=None):
        actions = np.ascontiguousarray(actions, np.int64)
        lens = np.ascontiguousarray(lens, np.int32)
        n, max_len = actions.shape
        idx_a = None if idx is None else np.ascontiguousarray(idx, np.int32)
        bufs, st = self._student(n)
        self._fn("reset_to_actions")(self.h, _p(actions), _p(lens), C.c_int32(max_len), _p(idx_a),
                                      C.c_int32(n), C.byref(st))
        return bufs

    def mutate_level(self, num_edits):
        bufs, st = self._student()
        self._fn("mutate_level")(self.h, C.c_int32(num_edits), C.byref(st))
        return bufs

    def step_buffers(self, term_init=None):
        B = self.B
        obs, obs_st = self._student()
        term, term_st = self._student()
        if term_init is not None:
            for k in term:
                term[k][...] = term_init[k]
        out = dict(obs=obs, term_obs=term, reward=np.zeros((B,), np.float32), done=np.zeros((B,), np.uint8),
                   info=np.zeros((B,), np.uint8), ep_ret=np.zeros((B,), np.float32),
                   ep_len=np.zeros((B,), np.int32), masks=np.zeros((B,), np.float32),
                   bad_masks=np.zeros((B,), np.float32), cliffhanger_masks=np.zeros((B,), np.float32))
        st = StepOut(obs_st, _p(out["reward"]), _p(out["done"]), _p(out["info"]), _p(out["ep_ret"]),
                     _p(out["ep_len"]), term_st, _p(out["masks"]), _p(out["bad_masks"]),
                     _p(out["cliffhanger_masks"]))
        return out, st

    def step(self, actions, flags=0, term_init=None):
        actions = np.ascontiguousarray(actions, np.int64)
        out, st = self.step_buffers(term_init)
        self._fn("step")(self.h, _p(actions), C.c_int32(flags), C.byref(st))
        return out

    def step_range(self, actions, flags, st, lo, hi):
        self._fn("step_range")(self.h, _p(actions), C.c_int32(flags), C.byref(st), C.c_int32(lo), C.c_int32(hi))

    def rollout_range(self, st, lo, hi, T, action_seed, t0, last_flag=1):
        """T fused steps of envs [lo,hi) with Philox actions (cpu_baseline leg; GIL released)."""
        return self._fn("rollout_range")(self.h, C.byref(st), C.c_int32(lo), C.c_int32(hi), C.c_int32(T),
                                         C.c_uint32(action_seed), C.c_uint32(t0), C.c_int32(last_flag))

    def get_encodings(self):
        out = np.zeros((self.B, self.S, self.S, 3), np.uint8)
        self._fn("get_encodings")(self.h, _p(out))
        return out

    def get_metrics(self):
        m = [np.zeros((self.B,), np.int32) for _ in range(4)]
        self._fn("get_metrics")(self.h, *[_p(a) for a in m])
        return dict(n_clutter_placed=m[0], passable=m[1], shortest_path_length=m[2], distance_to_goal=m[3])

    def get_state(self):
        out = np.zeros((self.B, 16), np.int32)
        self._fn("get_state")(self.h, _p(out))
        return out


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def random_actions(B, base, seed, t, num_actions):
    a = np.zeros((B,), np.int64)
    lib().orc_random_actions(_p(a), C.c_int32(B), C.c_int64(base), C.c_uint32(seed), C.c_uint32(t),
                             C.c_int32(num_actions))
    return a


def view_to_world(ax, ay, d, V, vx, vy):
    wx, wy = C.c_int(), C.c_int()
    lib().orc_view_to_world(ax, ay, d, V, vx, vy, C.byref(wx), C.byref(wy))
    return wx.value, wy.value


# ---- scans -----------------------------------------------------------------
def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def gae_returns(rewards, value_preds, masks, gamma, gae_lambda):
    r, v, m = _f32(rewards), _f32(value_preds), _f32(masks)
    T, B = r.shape
    out = np.zeros((T + 1, B), np.float32)
    lib().orc_gae_returns(_p(r), _p(v), _p(m), _p(out), C.c_int32(T), C.c_int32(B),
                          C.c_double(gamma), C.c_double(gae_lambda))
    return out


def discounted_returns(rewards, value_preds, masks, gamma):
    r, v, m = _f32(rewards), _f32(value_preds), _f32(masks)
    T, B = r.shape
    out = np.zeros((T + 1, B), np.float32)
    lib().orc_discounted_returns(_p(r), _p(v), _p(m), _p(out), C.c_int32(T), C.c_int32(B), C.c_double(gamma))
    return out


def batched_value_loss(returns, value_preds, mode=0, power=1, clipped=True):
    R, v = _f32(returns), _f32(value_preds)
    T, B = R.shape[0] - 1, R.shape[1]
    out = np.zeros((B,), np.float32)
    lib().orc_batched_value_loss(_p(R), _p(v), _p(out), C.c_int32(T), C.c_int32(B), C.c_int32(mode),
                                 C.c_int32(power), C.c_int32(int(clipped)))
    return out


def episode_scores(returns, value_preds, rewards, masks, cliffhanger_masks, level_seeds,
                   action_log_dist, strategy, gamma=0.995):
    R, v, r, m, c = map(_f32, (returns, value_preds, rewards, masks, cliffhanger_masks))
    seeds = np.ascontiguousarray(level_seeds, np.int32)
    T, B = r.shape
    if action_log_dist is not None:
        ld = _f32(action_log_dist)
        A = ld.shape[-1]
    else:
        ld, A = None, 0
    cap = B * (T + 1)
    recs = np.zeros((cap,), EPISODE_REC)
    n = lib().orc_episode_scores(_p(R), _p(v), _p(r), _p(m), _p(c), _p(seeds), _p(ld), C.c_int32(T),
                                 C.c_int32(B), C.c_int32(A), C.c_int32(strategy), C.c_double(gamma),
                                 _p(recs), C.c_int32(cap))
    return recs[:n].copy()


def grounded_scores(recs, grounded, value_preds, positive_only, dense):
    recs = np.ascontiguousarray(recs)
    g, v = _f32(grounded), _f32(value_preds)
    n = len(recs)
    mean, mx = np.zeros((n,), np.float32), np.zeros((n,), np.float32)
    lib().orc_grounded_scores(_p(recs), C.c_int32(n), _p(g), _p(v), C.c_int32(v.shape[1]),
                              C.c_int32(int(positive_only)), C.c_int32(int(dense)), _p(mean), _p(mx))
    return mean, mx


def rollout_return_stats(ep_ret, info):
    e = _f32(ep_ret)
    i = np.ascontiguousarray(info, np.uint8)
    T, B = e.shape
    mean, mx, n = np.zeros((B,), np.float32), np.zeros((B,), np.float32), np.zeros((B,), np.int32)
    lib().orc_rollout_return_stats(_p(e), _p(i), C.c_int32(T), C.c_int32(B), _p(mean), _p(mx), _p(n))
    return mean, mx, n


def env_return(agent_mean, agent_max, adv_mean, adv_max, mode=0):
    a, b, c, d = map(_f32, (agent_mean, agent_max, adv_mean, adv_max))
    out = np.zeros_like(a)
    lib().orc_env_return(_p(a), _p(b), _p(c), _p(d), _p(out), C.c_int32(a.shape[0]), C.c_int32(mode))
    return out


class VecNormalizeRet:
    """numpy restatement of VecNormalize(ob=False, ret=True) (envs/wrappers/vec_normalize.py:24-45,56-70) over baselines'
    RunningMeanStd (baselines/common/running_mean_std.py, third party, not in the reference tree: the standard
    parallel-variance merge with epsilon = 1e-4 — parity unpinned for that class)."""

    def __init__(self, num_envs, gamma=0.99, epsilon=1e-8, cliprew=10.0):
        self.ret = np.zeros(num_envs)
        self.mean, self.var, self.count = 0.0, 1.0, 1e-4
        self.gamma, self.epsilon, self.cliprew = gamma, epsilon, cliprew

    def reset(self):
        self.ret = np.zeros_like(self.ret)

    def step(self, rews, news):
        rews = np.asarray(rews, np.float32)
        self.ret = self.ret * self.gamma + rews
        bm, bv, bc = np.mean(self.ret), np.var(self.ret), self.ret.shape[0]
        delta, tot = bm - self.mean, self.count + bc
        m2 = self.var * self.count + bv * bc + np.square(delta) * self.count * bc / tot
        self.mean, self.var, self.count = self.mean + delta * bc / tot, m2 / tot, tot
        out = np.clip(rews / np.sqrt(self.var + self.epsilon), -self.cliprew, self.cliprew)
        self.ret[np.asarray(news, bool)] = 0.0
        return out


def bezier_curve(points, rad=0.2, edgy=0.2, numpoints=40):
    """envs/box2d/bezier.py:get_bezier_curve for one level: (x, y, a) with a = [n+1, 3] sorted closed control points + angles."""
    pts = np.ascontiguousarray(points, np.float64)
    n = pts.shape[0]
    xy = np.zeros((n * numpoints, 2), np.float64)
    ctrl = np.zeros((n + 1, 3), np.float64)
    m = lib().orc_bezier_curve(_p(pts), C.c_int32(n), C.c_double(rad), C.c_double(edgy), C.c_int32(numpoints), _p(xy), _p(ctrl))
    assert m == n * numpoints
    return xy[:, 0].copy(), xy[:, 1].copy(), ctrl
