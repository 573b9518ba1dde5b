"""HipAdversarialVecEnv — the reference's whole vec-env stack as ONE in-process GPU batch.

Replaces, with the same method set and return conventions:
  ParallelAdversarialVecEnv / SubprocVecEnv   envs/wrappers/parallel_wrappers.py:103-460
  TimeLimit (per env)                          envs/wrappers/time_limit.py:14-49
  VecMonitor                                   envs/wrappers/vec_monitor.py:18-85
  VecNormalize(ob=False, ret=False)            envs/wrappers/vec_normalize.py  (identity in MultiGrid configs)
  VecPreprocessImageWrapper(obs_key='image', transpose [2,0,1], scale 10)   envs/wrappers/obs_wrappers.py:40-230
Everything those layers compute per step is fused into dcd_mg_step (include/dcd_hip.h); this
class only converts arguments/results.

Two call levels:
  * reference-compatible (`step_env`, `step_adversary`, `reset_*`, getters): same signatures and
    result types as the reference — observation dicts of float tensors, reward FloatTensor[B,1],
    `done` numpy bool array, `infos` list of dicts with 'episode' / 'truncated' / 'truncated_obs'.
    Tensors live on the engine's device (the reference's are CPU tensors that the runner copies to
    the GPU, adversarial_runner.py:580).  Building `done`/`infos` synchronises the stream.
  * device level (`step_env_device`, `step_adversary_device`, ...): outputs go straight into
    caller-provided CUDA tensors (RolloutStorage slices), no synchronisation, no Python per env.
    envs.runners.AdversarialRunner uses this level.
"""
import numpy as np
import torch

import dcd_hip
from dcd_hip import INFO_CLIFFHANGER, INFO_EPISODE, INFO_TRUNC_KEY, INFO_TRUNCATED, STEP_LAST, STEP_RESET_RANDOM
from envs import spaces


class HipAdversarialVecEnv:
    def __init__(self, env_fns, adversary=True, is_eval=False, device="cuda:0", env_index_base=0, engine_seed=0,
                 normalize_returns=False, ret_gamma=0.99, cliprew=10.0, ret_epsilon=1e-8):
        if len(env_fns) == 0:
            raise ValueError("need at least one env_fn")
        self.closed = False
        self.proto = env_fns[0]()                    # level-family description (envs/multigrid/adversarial.py)
        self.num_envs = len(env_fns)
        self.spec = self.proto.spec
        self.is_eval = is_eval
        self.device = torch.device(device)
        self.env_index_base = int(env_index_base)
        cfg = self.proto.engine_config()
        # two replicas of the batch in one allocation: PAIRED's protagonist and antagonist play the teacher's levels in
        # independent rollouts (adversarial_runner.py:709-733); the second copy lets both advance in ONE launch per step
        self.engine = dcd_hip.MultiGridEngine(self.num_envs, device=device, engine_seed=int(engine_seed) & 0xFFFFFFFF,
                                              env_index_base=self.env_index_base,
                                              replicas=2 if cfg["size"] <= 15 else 1, **cfg)
        self._twin = None
        # VecNormalize(ob=False, ret=args.normalize_returns) of the reference's stack (util/__init__.py:193; defaults of
        # vec_normalize.py:24): running discounted returns + their RunningMeanStd live on the device, one kernel per step
        self.normalize_returns = bool(normalize_returns)
        self._ret_cfg = (float(ret_gamma), float(ret_epsilon), float(cliprew))
        if self.normalize_returns:
            self._ret = torch.zeros(self.num_envs, dtype=torch.float64, device=self.device)
            self.ret_rms = torch.tensor([0.0, 1.0, 1e-4], dtype=torch.float64, device=self.device)   # mean, var, count
        p = self.proto
        V, S = p.agent_view_size, p.size
        self.action_space = p.action_space
        # spaces as seen through VecPreprocessImageWrapper (channels first; obs_wrappers.py:117-146)
        self.observation_space = {"image": spaces.Box(0, 255, (3, V, V), "uint8"),
                                  "direction": p.observation_space["direction"]}
        self.is_fixed_level = bool(getattr(p, "is_fixed_level", False))
        # procedural families (mst_maze.py, cluttered.py): `_gen_grid` draws a NEW level on every reset in the reference,
        # so a level is regenerated on the host and re-uploaded whenever an episode ends (new_levels)
        self.is_procedural = bool(getattr(p, "is_procedural", False))
        if self.is_fixed_level:
            # evaluation mazes (envs/multigrid/maze.py): one description per env (corridor goals differ per env)
            self._descs = descs = [self.proto] + [fn() for fn in env_fns[1:]]
            if self.is_procedural:
                for i, d in enumerate(descs):                     # per-env generator streams (worker seeds, parallel_wrappers.py:414)
                    d.seed((int(engine_seed) * 1000003 + self.env_index_base + i) & 0x7FFFFFFF)
            self._fixed_levels = torch.from_numpy(np.stack([d.level_encoding() for d in descs])).to(self.device)
            self.adversary_action_space = None
            self.adversary_observation_space = None
        else:
            self.adversary_action_space = p.adversary_action_space
            self.adversary_observation_space = {"image": spaces.Box(0, 255, (3, S, S), "uint8"),
                                                "time_step": p.adversary_observation_space["time_step"],
                                                "random_z": p.adversary_observation_space["random_z"]}
        self.action_dim = 1
        self.adv_action_dim = 1 if adversary else 0
        self.processed_action_dim = 1
        self._seeds = list(range(self.env_index_base, self.env_index_base + self.num_envs))
        self._step_out = None
        self._adv_out = None

    # ------------------------------------------------------------------ helpers
    def _assert_not_closed(self):
        assert not self.closed, "Trying to operate on a HipAdversarialVecEnv after calling close()"

    def _actions(self, a):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(a)
        if not isinstance(a, torch.Tensor):
            a = torch.as_tensor(a)
        return a.to(device=self.device, dtype=torch.int64).reshape(self.num_envs).contiguous()

    @staticmethod
    def _student_dict(bufs):
        return {"image": bufs["image_f32"], "direction": bufs["dir_f32"]}

    @staticmethod
    def _adv_dict(bufs):
        return {"image": bufs["image_f32"], "time_step": bufs["time_step"], "random_z": bufs["random_z"]}

    def _student_out(self, n=None):
        return self.engine.student_buffers(n, u8=False, f32=True)

    def raise_errors(self):
        """Synchronises and raises the reference's exception for latched device-side errors."""
        self.engine.raise_errors()

    # ------------------------------------------------------------------ device level (no sync)
    def step_env_device(self, actions, out, reset_random=False, last_step=False):
        """actions: int64 CUDA tensor [B]; out: dict for MultiGridEngine.step (tensors may be storage slices)."""
        flags = (STEP_RESET_RANDOM if reset_random else 0) | (STEP_LAST if last_step else 0)
        out = self.engine.step(actions, flags, out)
        if self.normalize_returns:
            self._normalize(out["reward"], out["done"])
        return out

    def _normalize(self, reward, done):
        import ctypes as C
        from dcd_hip import check, lib, ptr, stream_ptr
        g, eps, clip = self._ret_cfg
        check(lib().dcd_return_normalize(ptr(reward, torch.float32, self.num_envs, "reward"), ptr(done, torch.uint8, self.num_envs, "done"),
                                         ptr(self._ret, torch.float64), ptr(self.ret_rms, torch.float64), C.c_double(g),
                                         C.c_double(eps), C.c_double(clip), C.c_int32(self.num_envs), stream_ptr()))

    def _reset_returns(self):
        if self.normalize_returns:                                    # vec_normalize.py:56-70: every reset* zeroes the running returns
            self._ret.zero_()

    @property
    def twin(self):
        """The second replica of the batch (dcd_mg_replica): the antagonist's copy of the levels."""
        if self._twin is None:
            self._twin = self.engine.replica(1)
        return self._twin

    def clone_to_twin(self):
        self.twin.clone_levels(self.engine)

    def reset_agent_twin_device(self, out):
        return self.twin.reset_agent(out)

    def step_env_pair_device(self, actions, twin_actions, out, twin_out, last_step=False):
        """One launch for both copies (dcd_mg_step_pair); no reset_random: PAIRED levels come from the teacher."""
        return self.engine.step_pair(self.twin, actions, twin_actions, STEP_LAST if last_step else 0, out, twin_out)

    def step_adversary_device(self, actions, out):
        return self.engine.step_adversary(actions, out)

    def reset_device(self, out):
        self._reset_returns()
        return self.engine.reset(out)

    def reset_agent_device(self, out):
        self._reset_returns()
        return self.engine.reset_agent(out)

    def reset_random_device(self, out):
        self._reset_returns()
        return self.engine.reset_random(out)

    # ------------------------------------------------------------------ reference-compatible level
    def reset(self):
        """Adversarial families: AdversarialEnv.reset (teacher obs).  Fixed-level families (eval mazes):
        MultiGridEnv.reset — reload the level, student obs (eval.py:266)."""
        self._assert_not_closed()
        self._reset_returns()
        if self.is_fixed_level:
            if self.is_procedural:
                self._fixed_levels.copy_(torch.from_numpy(np.stack([d.level_encoding() for d in self._descs])))
            out = self.engine.reset_to_fixed_level(self._fixed_levels, None, self._student_out())
            self.raise_errors()
            return self._student_dict(out)
        return self._adv_dict(self.engine.reset(self.engine.adv_buffers(u8=False)))

    def new_levels(self, idx):
        """Procedural families: the envs `idx` (list of ints) finished an episode — the worker's env.reset()
        (parallel_wrappers.py:20-25) draws a new level for each.  Returns their first observations (rows in idx order)."""
        idx = [int(i) for i in idx]
        enc = torch.from_numpy(np.stack([self._descs[i].level_encoding() for i in idx])).to(self.device)
        it = torch.tensor(idx, dtype=torch.int32, device=self.device)
        self._fixed_levels[it.long()] = enc
        out = self.engine.reset_to_fixed_level(enc, it, self._student_out(len(idx)))
        return self._student_dict(out)

    def step(self, actions):
        """VecEnv.step of the evaluation path (worker `step`, parallel_wrappers.py:20-25: env.reset() on done).
        For fixed levels reset() restores the same grid, which is what the fused auto-reset does."""
        return self.step_env(actions, reset_random=False)

    def reset_agent(self):
        self._assert_not_closed()
        self._reset_returns()
        obs = self._student_dict(self.engine.reset_agent(self._student_out()))
        self.raise_errors()
        return obs

    def reset_random(self):
        self._assert_not_closed()
        self._reset_returns()
        return self._student_dict(self.engine.reset_random(self._student_out()))

    def _levels_to_device(self, levels):
        """list of levels -> ('enc', uint8[n,S,S,3]) or ('actions', int64[n,L], int32[n])."""
        S = self.proto.size
        if isinstance(levels[0], str):
            seqs = [[int(a) for a in l.split()] for l in levels]
            L = max(1, max(len(s) for s in seqs))
            acts = np.zeros((len(seqs), L), np.int64)
            for i, s in enumerate(seqs):
                acts[i, :len(s)] = s
            lens = np.array([len(s) for s in seqs], np.int32)
            return "actions", torch.from_numpy(acts).to(self.device), torch.from_numpy(lens).to(self.device)
        enc = np.stack([np.frombuffer(l, dtype=np.uint8).reshape(S, S, 3) if isinstance(l, (bytes, bytearray))
                        else np.asarray(l, dtype=np.uint8).reshape(S, S, 3) for l in levels])
        return "enc", torch.from_numpy(np.ascontiguousarray(enc)).to(self.device), None

    def reset_to_level(self, level, index):
        """One env to a stored level (parallel_wrappers.py:341-347); returns obs with a leading dim of 1."""
        self._assert_not_closed()
        kind, a, b = self._levels_to_device([level])
        idx = torch.tensor([int(index)], dtype=torch.int32, device=self.device)
        out = self._student_out(1)
        if kind == "enc":
            self.engine.reset_to_encoding(a, idx, out)
        else:
            self.engine.reset_to_actions(a, b, idx, out)
        return self._student_dict(out)

    def reset_to_level_batch(self, level):
        self._assert_not_closed()
        if len(level) != self.num_envs:
            raise ValueError(f"reset_to_level_batch needs {self.num_envs} levels, got {len(level)}")
        kind, a, b = self._levels_to_device(list(level))
        out = self._student_out()
        if kind == "enc":
            self.engine.reset_to_encoding(a, None, out)
        else:
            self.engine.reset_to_actions(a, b, None, out)
        self.raise_errors()
        return self._student_dict(out)

    def reset_to_level_indices_device(self, encodings, indices, out):
        """Device level: uint8[n,S,S,3] encodings into env `indices` (int32[n]); obs rows into `out`."""
        return self.engine.reset_to_encoding(encodings, indices, out)

    def mutate_level(self, num_edits):
        self._assert_not_closed()
        out = self.engine.mutate_level(num_edits, self._student_out())
        self.raise_errors()
        return self._student_dict(out)

    def step_adversary(self, action):
        self._assert_not_closed()
        if self._adv_out is None:
            self._adv_out = self.engine.adv_buffers(u8=False)
        out = self.engine.step_adversary(self._actions(action), self._adv_out)
        self.raise_errors()
        obs = {k: v.clone() for k, v in self._adv_dict(out).items()}
        done = out["done"].cpu().numpy().astype(bool)
        reward = torch.zeros((self.num_envs, 1), dtype=torch.float32, device=self.device)
        return obs, reward, done, [dict() for _ in range(self.num_envs)]

    def step_env(self, actions, reset_random=False):
        """(obs, reward[B,1], done ndarray, infos) exactly as vec_env.py:122-127 through the wrapper stack."""
        self._assert_not_closed()
        if self._step_out is None:
            self._step_out = self.engine.step_buffers(u8=False)
        out = self.step_env_device(self._actions(actions), self._step_out, reset_random=reset_random)
        info = out["info"].cpu().numpy()                         # synchronises
        self.raise_errors()
        done = out["done"].cpu().numpy().astype(bool)
        ep_ret, ep_len = out["ep_ret"].cpu().numpy(), out["ep_len"].cpu().numpy()
        infos = []
        for i in range(self.num_envs):
            d = {}
            b = int(info[i])
            if b & INFO_TRUNC_KEY:
                d["truncated"] = bool(b & INFO_TRUNCATED)
                d["truncated_obs"] = {"image": out["term_obs"]["image_f32"][i:i + 1].clone(),
                                      "direction": out["term_obs"]["dir_f32"][i:i + 1].clone()}
            if b & INFO_EPISODE:
                d["episode"] = {"r": ep_ret[i], "l": int(ep_len[i]), "t": 0.0}
            infos.append(d)
        obs = {k: v.clone() for k, v in self._student_dict(out["obs"]).items()}
        if self.is_procedural and done.any():                     # env.reset() of a procedural family: a new level
            idx = np.nonzero(done)[0].tolist()
            first = self.new_levels(idx)
            for k in obs:
                obs[k][idx] = first[k]
            self.raise_errors()
        return obs, out["reward"].clone().unsqueeze(1), done, infos

    def step_env_async(self, action):
        self._pending = ("env", action, False)

    def step_env_reset_random_async(self, action):
        self._pending = ("env", action, True)

    def step_adversary_async(self, action):
        self._pending = ("adv", action, False)

    def step_wait(self):
        kind, action, rr = self._pending
        return self.step_adversary(action) if kind == "adv" else self.step_env(action, reset_random=rr)

    # ------------------------------------------------------------------ getters
    def get_observation_space(self):
        return self.observation_space

    def get_adversary_observation_space(self):
        return self.adversary_observation_space

    def get_adversary_action_space(self):
        return self.adversary_action_space

    def get_max_episode_steps(self):
        return self.proto._max_episode_steps

    def max_episode_steps(self):
        return self.proto._max_episode_steps

    @property
    def max_steps(self):
        return self.proto.max_steps

    def get_seed(self):
        return list(self._seeds)

    def set_seed(self, seeds):
        seeds = [int(s) for s in seeds]
        if len(seeds) != self.num_envs:
            raise ValueError(f"set_seed needs {self.num_envs} seeds")
        self._seeds = seeds
        self.engine.set_seed(torch.tensor(seeds, dtype=torch.int64))
        return [[s] for s in seeds]

    def seed(self, seed, index):
        self._seeds[index] = int(seed)
        self.engine.set_seed(torch.tensor(self._seeds, dtype=torch.int64))

    def get_encodings(self, index=None):
        enc = self.engine.get_encodings().cpu().numpy()
        if index is None or len(index) == 0:
            return [e for e in enc]
        return [enc[i] for i in index]

    def get_level(self):
        """venv.get_level() (parallel_wrappers.py:418-420: `[l[0] for l in remote_attr('level')]`; callers train.py:207 and
        adversarial_runner.py:376, both on the non-MultiGrid branch).  The reference's MultiGrid env defines no `level`
        attribute — there the call raises in the worker; here it returns each env's level in the form `reset_to_level`
        takes back (adversarial.py:271-294): the uint8[S,S,3] encoding, so get_level -> reset_to_level_batch round-trips."""
        return self.get_encodings()

    def get_encodings_device(self):
        return self.engine.get_encodings()

    def _metric(self, name):
        return self.engine.get_metrics()[name].cpu().numpy()

    def get_num_blocks(self):
        return self._metric("n_clutter_placed").tolist()

    def get_passable(self):
        p = self._metric("passable")
        return [(-1 if v < 0 else bool(v)) for v in p]

    def get_shortest_path_length(self):
        return self._metric("shortest_path_length").tolist()

    def get_distance_to_goal(self):
        return self._metric("distance_to_goal").tolist()

    def get_complexity_info(self):
        m = {k: v.cpu().numpy() for k, v in self.engine.get_metrics().items()}
        return [dict(num_blocks=int(m["n_clutter_placed"][i]), passable=int(m["passable"][i]),
                     shortest_path_length=int(m["shortest_path_length"][i])) for i in range(self.num_envs)]

    def remote_attr(self, name, data=None, flatten=False, index=None):
        table = {"encoding": self.get_encodings, "n_clutter_placed": self.get_num_blocks, "passable": self.get_passable,
                 "shortest_path_length": self.get_shortest_path_length, "distance_to_goal": self.get_distance_to_goal,
                 "seed_value": self.get_seed, "level": lambda: [[e] for e in self.get_encodings()]}
        if name == "seed" and data is not None:
            return self.set_seed(data)
        if name not in table:
            raise NotImplementedError(f"remote_attr('{name}') is not part of the MultiGrid hot path")
        vals = table[name]()
        if index is not None and len(index) > 0 and name != "encoding":
            vals = [vals[i] for i in index]
        return vals

    def get_images(self, index=None):
        """One RGB image per env of its current grid with the agent drawn in (parallel_wrappers.py:187-193 ->
        env.render('level'), multigrid.py:212-261): host-side, from the encodings and the agent state
        (envs/multigrid/rendering.py).  `index` limits the envs rendered (the reference renders all of them and
        the caller slices; at 4096 envs that is seconds of host work for one screenshot)."""
        from envs.multigrid.rendering import render_levels
        enc = self.engine.get_encodings().cpu().numpy()
        st = self.engine.get_state().cpu().numpy()
        index = range(len(enc)) if index is None else index
        out = []
        for i in index:
            e = enc[i].copy()
            ax, ay, ad = int(st[i, 0]), int(st[i, 1]), int(st[i, 2])
            if 0 <= ax < e.shape[0] and 0 <= ay < e.shape[1]:
                e[ax, ay] = (10, 0, ad)
            out.append(e)
        return render_levels(out) if out else []

    def render_to_screen(self):
        raise NotImplementedError("on-screen rendering is outside the GPU hot path")

    def close(self):
        if not self.closed:
            self.engine.close()
            self.closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
