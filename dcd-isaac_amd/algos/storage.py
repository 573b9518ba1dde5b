"""RolloutStorage — device-resident rollout buffers with the reference's fields and methods
(reference: algos/storage.py:41-517), the scans running as HIP kernels (dcd_hip.scans).

Layout is the reference's: time-major (T+1, B, ...) tensors with the env axis contiguous, which is
exactly what the one-env-per-lane scan kernels want.  Differences that do not change results:
  * buffers are allocated directly on `device` and the engine writes observations, rewards and masks
    straight into the [step] slices (`step_out()`), so there is no per-step host->device copy
    (reference: ~12 copy_ per step, :155-185);
  * `_compute_truncated_value_preds` forwards ALL (t, env) pairs with bad_masks == 0 in one batch
    instead of one net call per env (:208-231) — same inputs per row, fp32 GEMM batching tolerance;
  * the PPO minibatch generator gathers whole env trajectories with one index_select per tensor
    instead of a Python loop over envs (:466-480); the permutation draw is the reference's.
Extra device fields (`ep_ret`, `ep_info`, `dones`) keep per-step episode statistics so that the
runner's per-env return stats (adversarial_runner.py:257-271) need no host loop.
"""
from collections import defaultdict

import torch

from dcd_hip import scans


def _flatten(T, N, x):
    if isinstance(x, dict):
        return {k: v.reshape(T * N, *v.shape[2:]) for k, v in x.items()}
    return x.reshape(T * N, *x.shape[2:])


class LazyU8Obs(dict):
    """A PPO minibatch's student observations as the RAW bytes the rollout buffers hold (`image_u8` [R,V,V,3], `direction_u8` [R]):
    the fused feature kernel (models/fused_ppo.py) reads them directly — a quarter of the gather and no conversion pass.  The
    fp32 entries the reference's API exposes (`image` [R,3,V,V] = bytes / 10 in CHW, `direction` [R,1]; obs_wrappers.py:88-115)
    are materialised on first access by dcd_obs_u8_to_f32, bit-identical to the engine's own fp32 emission."""

    def __init__(self, image_u8, direction_u8, convert):
        super().__init__(image_u8=image_u8, direction_u8=direction_u8)
        self._convert = convert

    def __missing__(self, key):
        if key not in ("image", "direction"):
            raise KeyError(key)
        img8 = self["image_u8"]
        dst = {"image": torch.empty((img8.shape[0], 3, img8.shape[1], img8.shape[2]), device=img8.device),
               "direction": torch.empty((img8.shape[0], 1), device=img8.device)}
        self._convert({"image": img8, "direction": self["direction_u8"]}, dst, slice(None))
        dict.update(self, dst)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


class RolloutStorage:
    def __init__(self, model, num_steps, num_processes, observation_space, action_space,
                 recurrent_hidden_state_size, recurrent_arch="rnn", use_proper_time_limits=False,
                 use_popart=False, device="cpu"):
        self.device = torch.device(device)
        self.model = model
        self.num_steps, self.num_processes = num_steps, num_processes
        self.recurrent_arch = recurrent_arch
        self.recurrent_hidden_state_size = recurrent_hidden_state_size
        self.is_lstm = recurrent_arch == "lstm"
        self.use_proper_time_limits = use_proper_time_limits
        self.use_popart = bool(use_popart)      # value_preds are normalised; model.popart denormalises them (:240-247)
        T, B, d = num_steps, num_processes, self.device
        z = lambda *s, **kw: torch.zeros(*s, device=d, **kw)
        self.is_dict_obs = isinstance(observation_space, dict)
        self.truncated_obs = None
        self.obs_u8 = self.truncated_obs_u8 = None       # u8 mode (enable_u8_obs): what the engine writes; fp32 derived
        self._stale = None
        if self.is_dict_obs:
            self._obs = {k: z(T + 1, B, *sp.shape) for k, sp in observation_space.items()}
            if use_proper_time_limits:
                self.truncated_obs = {k: z(T + 1, B, *sp.shape) for k, sp in observation_space.items()}
        else:
            self._obs = z(T + 1, B, *observation_space.shape)
            if use_proper_time_limits:
                self.truncated_obs = torch.zeros_like(self._obs)
        hs = 2 * recurrent_hidden_state_size if self.is_lstm else recurrent_hidden_state_size
        self.recurrent_hidden_states = z(T + 1, B, hs)
        self.rewards = z(T, B, 1)
        self.value_preds = z(T + 1, B, 1)
        self.returns = z(T + 1, B, 1)
        self.action_log_probs = z(T, B, 1)
        if action_space.__class__.__name__ == "Discrete":
            action_shape = 1
            self.action_log_dist = z(T, B, action_space.n)
        else:
            action_shape = action_space.shape[0]
            self.action_log_dist = z(T, B, 1)
        self.actions = z(T, B, action_shape)
        if action_space.__class__.__name__ == "Discrete":
            self.actions = self.actions.long()
        self.masks = torch.ones(T + 1, B, 1, device=d)
        self.bad_masks = torch.ones(T + 1, B, 1, device=d)
        self.cliffhanger_masks = torch.ones(T + 1, B, 1, device=d)
        self.truncated_value_preds = torch.zeros_like(self.value_preds) if use_proper_time_limits else None
        self.denorm_value_preds = None
        self.level_seeds = z(T, B, 1, dtype=torch.int)
        # device-side episode statistics of the current rollout (info bits: dcd_hip.INFO_*)
        self.ep_ret = z(T, B)
        self.ep_info = z(T, B, dtype=torch.uint8)
        self.dones = z(T, B, dtype=torch.uint8)
        self.step = 0

    def to(self, device):
        device = torch.device(device)
        self.device = device
        for name, val in list(vars(self).items()):
            if isinstance(val, torch.Tensor):
                setattr(self, name, val.to(device))
            elif isinstance(val, dict) and val and all(isinstance(v, torch.Tensor) for v in val.values()):
                setattr(self, name, {k: v.to(device) for k, v in val.items()})
        return self

    # ---- observation access ---------------------------------------------------------------
    # u8 mode (student storages of device rollouts): the engine writes the RAW observation — the encoding bytes
    # [vx][vy][c] and the direction byte, 76 B per env-step instead of the 304 B of the preprocessed fp32 CHW image — into
    # `obs_u8` / `truncated_obs_u8`; the fused policy step reads those bytes directly (dcd_policy_features_u8).  The
    # reference's fp32 tensors (`obs`, what VecPreprocessImageWrapper hands over, obs_wrappers.py:88-115) are DERIVED:
    # rows are converted on demand (one step for `get_obs`, the whole rollout once per PPO update) by
    # dcd_obs_u8_to_f32, bit-identical to what the engine's own fp32 emission writes.
    def enable_u8_obs(self):
        """Students: image + direction as bytes.  The teacher (image, time_step, random_z): the full-grid image as bytes
        (675 instead of 2 700 B per env-step), the two small float entries stay as they are."""
        if self.obs_u8 is not None or not self.is_dict_obs or "image" not in self._obs or self._obs["image"].dim() != 5:
            return self.obs_u8 is not None
        T1, B, d = self.num_steps + 1, self.num_processes, self.device
        V = self._obs["image"].shape[-1]
        has_dir = "direction" in self._obs

        def mk():
            o = {"image": torch.zeros((T1, B, V, V, 3), dtype=torch.uint8, device=d)}
            if has_dir:
                o["direction"] = torch.zeros((T1, B), dtype=torch.uint8, device=d)
            return o
        self.obs_u8 = mk()
        self._stale = [False] * T1
        if self.truncated_obs is not None:
            self.truncated_obs_u8 = mk()
        return True

    def _convert(self, src, dst, sl):
        import ctypes as C
        from dcd_hip import check, lib, ptr, stream_ptr
        img8 = src["image"][sl]
        n, V = img8.numel() // (img8.shape[-1] * img8.shape[-2] * img8.shape[-3]), img8.shape[-2]
        has_dir = "direction" in src
        check(lib().dcd_obs_u8_to_f32(ptr(img8, torch.uint8), ptr(src["direction"][sl], torch.uint8) if has_dir else None,
                                      ptr(dst["image"][sl], torch.float32),
                                      ptr(dst["direction"][sl], torch.float32) if has_dir else None, C.c_int64(n), C.c_int32(V), stream_ptr()))

    def mark_obs_stale(self):
        """Every fp32 row is out of date (a HIP-graph replay rewrote the u8 buffers without passing through Python)."""
        if self.obs_u8 is not None:
            self._stale = [True] * (self.num_steps + 1)

    def sync_obs(self, idx=None):
        """Brings the fp32 observation rows up to date with the u8 ones (all stale time steps, or just `idx`)."""
        if self.obs_u8 is None:
            return
        if idx is not None:
            idx = idx % (self.num_steps + 1)
            if self._stale[idx]:
                self._convert(self.obs_u8, self._obs, slice(idx, idx + 1))
                self._stale[idx] = False
            return
        t, T1 = 0, self.num_steps + 1
        while t < T1:                                     # contiguous runs of stale steps, one launch each
            if not self._stale[t]:
                t += 1
                continue
            e = t
            while e < T1 and self._stale[e]:
                self._stale[e] = False
                e += 1
            self._convert(self.obs_u8, self._obs, slice(t, e))
            t = e

    @property
    def obs(self):
        self.sync_obs()
        return self._obs

    def get_obs(self, idx):
        self.sync_obs(idx)
        return {k: v[idx] for k, v in self._obs.items()} if self.is_dict_obs else self._obs[idx]

    def copy_obs_to_index(self, obs, index):
        if self.obs_u8 is not None:
            raise RuntimeError("storage in u8 observation mode: fp32 observations are derived; write obs_u8 (engine outputs)")
        if self.is_dict_obs:
            for k in self._obs:
                self._obs[k][index].copy_(obs[k])
        else:
            self._obs[index].copy_(obs)

    def student_obs_out(self, index, truncated=False):
        """Engine output struct whose image/direction land in obs[index] (or truncated_obs[index]): raw bytes in u8 mode,
        the preprocessed fp32 otherwise."""
        if self.obs_u8 is not None:
            src = self.truncated_obs_u8 if truncated else self.obs_u8
            if not truncated:
                self._stale[index % (self.num_steps + 1)] = True
            return {"image_u8": src["image"][index], "dir_u8": src["direction"][index]}
        src = self.truncated_obs if truncated else self._obs
        return {"image_f32": src["image"][index], "dir_f32": src["direction"][index]}

    def adversary_obs_out(self, index):
        if self.obs_u8 is not None:
            self._stale[index % (self.num_steps + 1)] = True
            return {"image_u8": self.obs_u8["image"][index], "time_step": self._obs["time_step"][index],
                    "random_z": self._obs["random_z"][index]}
        return {"image_f32": self._obs["image"][index], "time_step": self._obs["time_step"][index],
                "random_z": self._obs["random_z"][index]}

    def rollout_obs(self, idx):
        """Observation of step `idx` as the rollout-time HIP policy paths read it: the image as raw bytes in u8 mode (no
        conversion), everything else as stored."""
        if self.obs_u8 is None:
            return self.get_obs(idx)
        o = {k: v[idx] for k, v in self._obs.items() if k not in self.obs_u8}
        o.update({k: v[idx] for k, v in self.obs_u8.items()})
        return o

    def step_out(self, step):
        """DcdStepOut for rollout step `step`: the engine writes obs[step+1], rewards[step], masks[step+1],
        bad_masks[step+1], cliffhanger_masks[step+1], truncated_obs[step+1] and the episode stats
        (what adversarial_runner.py:517-585 + storage.insert do on the host in the reference)."""
        out = dict(obs=self.student_obs_out(step + 1), reward=self.rewards[step].view(-1),
                   done=self.dones[step], info=self.ep_info[step], ep_ret=self.ep_ret[step],
                   masks=self.masks[step + 1].view(-1), bad_masks=self.bad_masks[step + 1].view(-1),
                   cliffhanger_masks=self.cliffhanger_masks[step + 1].view(-1))
        if self.truncated_obs is not None:
            out["term_obs"] = self.student_obs_out(step + 1, truncated=True)
        return out

    # ---- insert (reference :155-203) --------------------------------------------------------
    def insert(self, obs, recurrent_hidden_states, actions, action_log_probs, action_log_dist, value_preds,
               rewards, masks, bad_masks, level_seeds=None, cliffhanger_masks=None):
        if rewards.dim() == 3:
            rewards = rewards.squeeze(2)
        self.copy_obs_to_index(obs, self.step + 1)
        self.insert_policy(recurrent_hidden_states, actions, action_log_probs, action_log_dist, value_preds,
                           level_seeds)
        self.rewards[self.step].copy_(rewards)
        self.masks[self.step + 1].copy_(masks)
        self.bad_masks[self.step + 1].copy_(bad_masks)
        if cliffhanger_masks is not None:
            self.cliffhanger_masks[self.step + 1].copy_(cliffhanger_masks)
        self.advance()

    def insert_policy(self, recurrent_hidden_states, actions, action_log_probs, action_log_dist, value_preds,
                      level_seeds=None):
        """Policy-side half of insert(); the env-side half is written by the engine via step_out()."""
        s, H = self.step, self.recurrent_hidden_state_size
        if self.is_lstm:
            self.recurrent_hidden_states[s + 1, :, :H].copy_(recurrent_hidden_states[0])
            self.recurrent_hidden_states[s + 1, :, H:].copy_(recurrent_hidden_states[1])
        else:
            self.recurrent_hidden_states[s + 1].copy_(recurrent_hidden_states)
        self.actions[s].copy_(actions)
        self.action_log_probs[s].copy_(action_log_probs)
        self.action_log_dist[s].copy_(action_log_dist)
        self.value_preds[s].copy_(value_preds)
        if level_seeds is not None:
            self.level_seeds[s].copy_(level_seeds)

    def advance(self):
        self.step = (self.step + 1) % self.num_steps

    def insert_truncated_obs(self, obs, index):
        if self.truncated_obs_u8 is not None:
            raise RuntimeError("storage in u8 observation mode: truncated observations come from the engine")
        for k in self.truncated_obs:
            self.truncated_obs[k][self.step + 1][index].copy_(torch.as_tensor(obs[k]).reshape(
                self.truncated_obs[k].shape[2:]))

    def after_update(self):
        if self.obs_u8 is not None:
            for k in self.obs_u8:
                self.obs_u8[k][0].copy_(self.obs_u8[k][-1])
            for k in self._obs:
                if k not in self.obs_u8:                                  # entries that are not kept as bytes (teacher: time_step, random_z)
                    self._obs[k][0].copy_(self._obs[k][-1])
            self._stale[0] = True
        elif self.is_dict_obs:
            for k in self._obs:
                self._obs[k][0].copy_(self._obs[k][-1])
        else:
            self._obs[0].copy_(self._obs[-1])
        self.recurrent_hidden_states[0].copy_(self.recurrent_hidden_states[-1])
        self.masks[0].copy_(self.masks[-1])
        self.bad_masks[0].copy_(self.bad_masks[-1])
        self.cliffhanger_masks[0].copy_(self.cliffhanger_masks[-1])

    def replace_final_return(self, returns):
        self.rewards[-1] = returns

    # ---- returns --------------------------------------------------------------------------
    def _compute_truncated_value_preds(self):
        """value_preds with V(truncated_obs) written wherever bad_masks == 0 (reference :208-231)."""
        self.truncated_value_preds.copy_(self.value_preds)
        bad = self.bad_masks[:, :, 0] == 0
        # Reference quirk (:214-216): `steps = nonzero().squeeze()` is 0-dim when an env has exactly ONE
        # such step and that env is then skipped — only envs with >= 2 truncated steps are forwarded.
        bad = bad & (bad.sum(dim=0, keepdim=True) >= 2)
        t_idx, b_idx = bad.nonzero(as_tuple=True)
        if t_idx.numel() == 0:
            return self.truncated_value_preds
        with torch.no_grad():
            if self.truncated_obs_u8 is not None:         # gather the raw rows, preprocess just those
                g8 = {k: v[t_idx, b_idx].contiguous() for k, v in self.truncated_obs_u8.items()}
                n = t_idx.numel()
                obs = {"image": torch.empty((n,) + tuple(self._obs["image"].shape[2:]), device=self.device),
                       "direction": torch.empty((n, 1), device=self.device)}
                self._convert(g8, obs, slice(None))
            elif self.is_dict_obs:
                obs = {k: v[t_idx, b_idx] for k, v in self.truncated_obs.items()}
            else:
                obs = self.truncated_obs[t_idx, b_idx]
            hxs = self.recurrent_hidden_states[t_idx, b_idx]
            if self.is_lstm:
                hxs = self._split_batched_lstm_recurrent_hidden_states(hxs)
            ones = torch.ones((t_idx.numel(), 1), device=self.device)
            self.truncated_value_preds[t_idx, b_idx] = self.model.get_value(obs, hxs, ones)
        return self.truncated_value_preds

    def _value_source(self, next_value):
        if next_value.data_ptr() != self.value_preds[-1].data_ptr():      # already in place (fused bootstrap value)
            self.value_preds[-1] = next_value
        v = self._compute_truncated_value_preds() if self.use_proper_time_limits else self.value_preds
        if self.use_popart:                          # storage.py:247-249,269-270: every value prediction denormalised
            self.denorm_value_preds = self.model.popart.denormalize(v).contiguous()
        return v

    def compute_gae_returns(self, returns_buffer, next_value, gamma, gae_lambda):
        v = self._value_source(next_value)
        scans.gae_returns(self.rewards, self.denorm_value_preds if self.use_popart else v, self.masks, returns_buffer, gamma, gae_lambda)

    def compute_discounted_returns(self, returns_buffer, next_value, gamma):
        # the reference seeds the scan with the NORMALISED last value even under PopArt (:266-272: denorm_value_preds is
        # computed but returns[-1] = value_preds[-1]); kept
        scans.discounted_returns(self.rewards, self._value_source(next_value), self.masks, returns_buffer, gamma)

    @staticmethod
    def compute_returns_batch(storages, next_values, use_gae, gamma, gae_lambda, stat_sets=None):
        """compute_returns of several storages of one shape (PAIRED: protagonist and antagonist) with ONE scan launch;
        stat_sets (rollout_return_stats_batch's argument): the rollouts' return statistics ride in the same launch."""
        if not use_gae or len(storages) == 1 or len({(s.num_steps, s.num_processes) for s in storages}) != 1:
            for s, nv in zip(storages, next_values):
                s.compute_returns(nv, use_gae, gamma, gae_lambda)
            if stat_sets is not None:
                scans.rollout_return_stats_batch(stat_sets)
            return
        sets = []
        for s, nv in zip(storages, next_values):
            v = s._value_source(nv)
            sets.append((s.rewards, s.denorm_value_preds if s.use_popart else v, s.masks, s.returns))
        if stat_sets is not None:
            scans.rollout_tail_batch(sets, stat_sets, gamma, gae_lambda)
        else:
            scans.gae_returns_batch(sets, gamma, gae_lambda)

    def compute_returns(self, next_value, use_gae, gamma, gae_lambda):
        if use_gae:
            self.compute_gae_returns(self.returns, next_value, gamma, gae_lambda)
        else:
            self.compute_discounted_returns(self.returns, next_value, gamma)

    def get_batched_value_loss(self, signed=False, positive_only=False, power=1, clipped=True, batched=True):
        mode = 1 if signed else (2 if positive_only else 0)
        v = self.denorm_value_preds if self.use_popart else self.value_preds      # :301-306
        out = scans.batched_value_loss(self.returns, v, mode, int(power), clipped)
        return out if batched else out.mean().item()

    # ---- action trajectories ----------------------------------------------------------------
    def get_action_traj(self, as_string=False):
        if as_string:
            acts = self.actions[:, :, 0].t().cpu().numpy()
            digits = [str(i) for i in range(int(acts.max()) + 1)] if acts.size and acts.min() >= 0 else None
            if digits is None:
                return [" ".join(map(str, row)) for row in acts.tolist()]
            return [" ".join(map(digits.__getitem__, row)) for row in acts.tolist()]      # 20 -> 5 ms at 4096 levels
        return self.actions.squeeze(-1)

    def _episode_action_strings(self):
        acts = self.actions[:, :, 0].cpu().numpy()
        masks = self.masks[:, :, 0].cpu().numpy()
        for b in range(acts.shape[1]):
            done_steps = [0] + [int(t) for t in (masks[:, b] == 0).nonzero()[0]]
            for i, t in enumerate(done_steps[:-1]):
                yield b, " ".join(str(a) for a in acts[t:done_steps[i + 1], b])

    def action_complexity_inputs(self):
        """Host copies of what the action complexity reads (actions, masks): taken when the rollout is complete, so that the
        LZ76 pass itself can run later, beside the PPO updates (which rewrite masks[0] in after_update)."""
        import numpy as np
        return (np.ascontiguousarray(self.actions[:, :, 0].cpu().numpy().astype(np.int64)),
                np.ascontiguousarray(self.masks[:, :, 0].cpu().numpy().astype(np.float32)))

    def _action_complexity(self, per_env, inputs=None):
        """dcd_action_complexity (csrc/host_utils.cpp): the per-character LZ76 loop in C, on host copies of the
        action and mask buffers (1M characters per rollout at 4096 envs: 0.37 s in Python)."""
        import ctypes as C
        import numpy as np
        from dcd_hip import check, lib
        acts, masks = inputs if inputs is not None else self.action_complexity_inputs()
        T, B = acts.shape
        total, n = C.c_double(0), C.c_int64(0)
        env = np.zeros(B, np.float32) if per_env else None
        check(lib().dcd_action_complexity(acts.ctypes.data_as(C.c_void_p), masks.ctypes.data_as(C.c_void_p), C.c_int32(T),
                                          C.c_int32(B), C.byref(total), C.byref(n),
                                          env.ctypes.data_as(C.c_void_p) if per_env else None))
        return total.value, n.value, env

    def get_action_complexity(self, inputs=None):
        total, n, _ = self._action_complexity(False, inputs)
        return total / max(n, 1)

    def get_batched_action_complexity(self):
        _, _, env = self._action_complexity(True)
        return torch.from_numpy(env).view(-1, 1)

    # ---- recurrent state helpers ----------------------------------------------------------------
    def _split_batched_lstm_recurrent_hidden_states(self, hxs):
        H = self.recurrent_hidden_state_size
        return hxs[:, :H], hxs[:, H:]

    def get_recurrent_hidden_state(self, step):
        if self.is_lstm:
            return self._split_batched_lstm_recurrent_hidden_states(self.recurrent_hidden_states[step])
        return self.recurrent_hidden_states[step]

    # ---- PPO minibatch generators (reference :400-517) ------------------------------------------
    def predraw_permutations(self, n_epochs, recurrent):
        """Draws the minibatch permutations of the next `n_epochs` PPO epochs NOW, on the calling thread, from the global
        torch generator (the draws torch.randperm would make inside the generators, in the same order).  The runner calls
        it for each queued update in program order before running the updates concurrently on host threads, so that a
        fixed --seed gives the same minibatches whatever the thread scheduling."""
        n = self.num_processes if recurrent else self.rewards.shape[0] * self.rewards.shape[1]
        self._perms = [torch.randperm(n) for _ in range(n_epochs)]

    def _next_perm(self, n):
        perms = getattr(self, "_perms", None)
        if perms:
            p = perms.pop(0)
            assert p.numel() == n
            return p
        return torch.randperm(n)

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None):
        T, B = self.rewards.shape[:2]
        total = T * B
        if mini_batch_size is None:
            assert total >= num_mini_batch
            mini_batch_size = total // num_mini_batch
        perm = self._next_perm(total)
        for s in range(0, total, mini_batch_size):
            idx = perm[s:s + mini_batch_size].to(self.device)
            flat = lambda x: x.reshape(-1, *x.shape[2:])[idx]
            obs = {k: flat(v[:-1]) for k, v in self.obs.items()} if self.is_dict_obs else flat(self.obs[:-1])
            hxs = flat(self.recurrent_hidden_states[:-1])
            if self.is_lstm:
                hxs = self._split_batched_lstm_recurrent_hidden_states(hxs)
            yield (obs, hxs, flat(self.actions), flat(self.value_preds[:-1]), flat(self.returns[:-1]),
                   flat(self.masks[:-1]), flat(self.action_log_probs),
                   None if advantages is None else flat(advantages))

    def recurrent_generator(self, advantages, num_mini_batch):
        B, T = self.num_processes, self.num_steps
        assert B >= num_mini_batch, (
            f"PPO requires the number of processes ({B}) to be greater than or equal to "
            f"the number of PPO mini batches ({num_mini_batch}).")
        per = B // num_mini_batch
        perm = self._next_perm(B)                                    # same draw as the reference (:450)
        for s in range(0, B, per):
            idx = perm[s:s + per].to(self.device)
            N = idx.numel()
            sel = lambda x: x.index_select(1, idx)
            if self.obs_u8 is not None and set(self.obs_u8) == {"image", "direction"} and set(self._obs) == {"image", "direction"}:
                # students in u8 mode: the minibatch carries the raw bytes; fp32 views on demand (LazyU8Obs)
                i8, d8 = sel(self.obs_u8["image"][:-1]), sel(self.obs_u8["direction"][:-1])
                obs_b = LazyU8Obs(i8.reshape(T * N, *i8.shape[2:]), d8.reshape(T * N), self._convert)
            else:
                obs = ({k: sel(v[:-1]) for k, v in self.obs.items()} if self.is_dict_obs else sel(self.obs[:-1]))
                obs_b = _flatten(T, N, obs)
            hxs = self.recurrent_hidden_states[0].index_select(0, idx)
            if self.is_lstm:
                hxs = self._split_batched_lstm_recurrent_hidden_states(hxs)
            yield (obs_b, hxs, _flatten(T, N, sel(self.actions)),
                   _flatten(T, N, sel(self.value_preds[:-1])), _flatten(T, N, sel(self.returns[:-1])),
                   _flatten(T, N, sel(self.masks[:-1])), _flatten(T, N, sel(self.action_log_probs)),
                   _flatten(T, N, sel(advantages)))


def lempel_ziv_complexity(sequence):
    """Number of phrases in the LZ76 parsing of `sequence` (what the `lempel_ziv_complexity` package
    used by the reference's action-complexity logging computes, storage.py:21,329-369)."""
    phrases, i, n = set(), 0, len(sequence)
    while i < n:
        j = i + 1
        while j <= n and sequence[i:j] in phrases:
            j += 1
        phrases.add(sequence[i:j])
        i = j
    return len(phrases)
