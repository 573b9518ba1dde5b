"""LevelSampler — Prioritized Level Replay bookkeeping with the reference's constructor, attributes
and method set (reference: level_replay/level_sampler.py:19-800).

Split of work:
  * segmentation of the rollout into per-(actor, episode) records and the per-record score
    (mean/max of the strategy's per-step quantity) run on the GPU: dcd_hip.scans.EpisodeScorer
    (kernels seg_count / seg_scan / seg_score in csrc/scan_kernels.hip), emitted in the reference's
    iteration order (actor-major, then time; :497-578);
  * the order-dependent bookkeeping (partial-score merging, exponential averaging, staging ->
    working buffer insertion, staleness, sampling weights) stays on the host in float64 numpy
    exactly as in the reference, consuming the records sequentially.  `after_update`, an
    O(num_actors x buffer) pure-Python double loop in the reference (:585-588), visits only the
    non-zero partial scores, in the same order.

`update_with_records` is the host half on its own (used by the CPU tests with oracle records and by
the multi-GPU path, where rank 0 consumes the all-gathered records of every rank).
"""
from collections import defaultdict

import numpy as np

INT32_MAX = 2147483647

# strategy -> (dcd_episode_scores code, needs returns/values)
_GPU_STRATEGY = {"uniform": 7, "policy_entropy": 3, "least_confidence": 4, "min_margin": 5, "gae": 2,
                 "value_l1": 0, "signed_value_loss": 2, "positive_value_loss": 1,
                 "grounded_signed_value_loss": 8, "grounded_positive_value_loss": 8, "one_step_td_error": 6}
_VALUE_STRATEGIES = ("gae", "value_l1", "signed_value_loss", "positive_value_loss", "grounded_signed_value_loss",
                     "grounded_positive_value_loss", "one_step_td_error", "alt_advantage_abs", "tscl_window")


def _scoring_values(rollouts):
    """Value predictions the score functions compare returns with: denormalised under PopArt (level_sampler.py:520-525)."""
    return rollouts.denorm_value_preds if getattr(rollouts, "use_popart", False) else rollouts.value_preds


_NATIVE_DRAWS = None


def _native_draws_available():
    """True when libdcd_hip's dcd_sampler_draws reproduces the numpy draw loop bit for bit on this machine (checked once
    on a random buffer: indices and final staleness must be identical); DCD_NATIVE_SAMPLER=0 disables it."""
    global _NATIVE_DRAWS
    if _NATIVE_DRAWS is None:
        import os
        _NATIVE_DRAWS = False
        if os.environ.get("DCD_NATIVE_SAMPLER", "1") != "0":
            try:
                import ctypes as C
                from dcd_hip import lib
                L = lib()
                L.dcd_sampler_draws.restype = C.c_int64
                rng = np.random.RandomState(12345)
                N, n, c = 257, 64, 0.3
                wsc = (1 - c) * (lambda v: v / v.sum())(rng.rand(N))
                keep = (rng.rand(N) > 0.2).astype(np.float64)
                st0 = rng.randint(0, 30, N).astype(np.float64)
                u = rng.random_sample(n)
                st, idx = st0.copy(), np.empty(n, dtype=np.int64)
                p_ = lambda a: a.ctypes.data_as(C.c_void_p)
                k = L.dcd_sampler_draws(C.c_int64(N), C.c_int64(n), p_(wsc), C.c_double(c), p_(st), p_(keep), p_(u), C.c_int32(1), p_(idx))
                ref_st, ref_idx = st0.copy(), []
                for j in range(n):
                    sw = ref_st * keep
                    sw /= np.sum(sw)
                    w = wsc + c * sw
                    total = np.sum(w)
                    if total != 1.0:
                        w = w / total
                    cdf = w.cumsum()
                    cdf /= cdf[-1]
                    i = int(cdf.searchsorted(u[j:j + 1], side="right")[0])
                    ref_idx.append(i)
                    ref_st = ref_st + 1
                    ref_st[i] = 0
                _NATIVE_DRAWS = bool(k == n and idx.tolist() == ref_idx and np.array_equal(st, ref_st))
            except Exception:
                _NATIVE_DRAWS = False
    return _NATIVE_DRAWS


class _RecordingRng:
    """A RandomState that remembers the doubles `random_sample` handed out (the speculated draws' uniforms); any other way of
    consuming the stream marks the recording unsafe."""

    def __init__(self, rs):
        self.rs, self.doubles, self.unsafe = rs, [], False

    def random_sample(self, n):
        d = self.rs.random_sample(n)
        self.doubles.append(np.array(d, copy=True))
        return d

    def rand(self, *a):
        self.unsafe = True
        return self.rs.rand(*a)

    def choice(self, *a, **k):
        self.unsafe = True
        return self.rs.choice(*a, **k)

    def get_state(self):
        return self.rs.get_state()


class LevelSampler:
    def __init__(self, seeds, obs_space, action_space, num_actors=1, strategy="random", max_score_coef=0.0,
                 replay_schedule="fixed", score_transform="power", temperature=1.0, eps=0.05, rho=1.0,
                 replay_prob=0.95, alpha=1.0, staleness_coef=0, staleness_transform="power",
                 staleness_temperature=1.0, sample_full_distribution=False, seed_buffer_size=0,
                 seed_buffer_priority="replay_support", use_dense_rewards=False, tscl_window_size=0, gamma=0.999):
        if strategy in ("tscl_window", "alt_advantage_abs"):
            raise NotImplementedError(f"strategy '{strategy}' is outside the MultiGrid configs and not supported")
        self.obs_space, self.action_space = obs_space, action_space
        self.num_actors = num_actors
        self.strategy = strategy
        self.max_score_coef = max_score_coef
        self.replay_schedule = replay_schedule
        self.score_transform = score_transform
        self.temperature = temperature
        self.eps = eps
        self.rho = rho
        self.replay_prob = replay_prob
        self.alpha = alpha
        self.staleness_coef = staleness_coef
        self.staleness_transform = staleness_transform
        self.staleness_temperature = staleness_temperature
        self.gamma = gamma
        self.use_dense_rewards = use_dense_rewards

        self.seed_buffer_size = seed_buffer_size if not seeds else len(seeds)
        N = self.seed_buffer_size
        self._init_seed_index(seeds)
        self.unseen_seed_weights = np.ones(N, dtype=np.float64)
        self.seed_scores = np.zeros(N, dtype=np.float64)
        self.partial_seed_scores = np.zeros((num_actors, N), dtype=np.float64)
        self.partial_seed_max_scores = np.full((num_actors, N), -np.inf, dtype=np.float64)
        self.partial_seed_steps = np.zeros((num_actors, N), dtype=np.int32)
        self.seed_staleness = np.zeros(N, dtype=np.float64)
        self.running_sample_count = 0
        self.next_seed_index = 0
        self.track_solvable = False
        self.grounded_values = np.full(N, -np.inf, dtype=np.float64) if strategy.startswith("grounded") else None

        self.sample_full_distribution = sample_full_distribution
        if sample_full_distribution:
            self.seed2actor = defaultdict(set)
            self.working_seed_buffer_size = 0
            self.seed_buffer_priority = seed_buffer_priority
            self.staging_seed_set = set()
            self.working_seed_set = set()
            self.seed2timestamp_buffer = {}
            self.partial_seed_scores_buffer = [{} for _ in range(num_actors)]
            self.partial_seed_max_scores_buffer = [{} for _ in range(num_actors)]
            self.partial_seed_steps_buffer = [{} for _ in range(num_actors)]
        self._scorer = None
        self._state_version, self._weakest_memo = 0, None
        # (actor, seed index) pairs holding a partial score: after_update visits exactly these instead of scanning the
        # whole [num_actors, N] matrix (1 GB at 32 768 actors x 4000 seeds)
        self._partial_dirty = set()

    # ---- seed index ---------------------------------------------------------------------------
    def seed_range(self):
        if not self.sample_full_distribution:
            return int(min(self.seeds)), int(max(self.seeds))
        return 0, INT32_MAX

    def _init_seed_index(self, seeds):
        if seeds:
            self.seeds = np.array(seeds, dtype=np.int64)
            self.seed2index = {seed: i for i, seed in enumerate(seeds)}
        else:
            self.seeds = np.zeros(self.seed_buffer_size, dtype=np.int64) - 1
            self.seed2index = {}

    def _init_solvable_tracking(self):
        self.track_solvable = True
        self.staging_seed2solvable = {}
        self.seed_solvable = np.ones(self.seed_buffer_size, dtype=bool)

    @property
    def _proportion_filled(self):
        if self.sample_full_distribution:
            return self.working_seed_buffer_size / self.seed_buffer_size
        num_unseen = (self.unseen_seed_weights > 0).sum()
        return (len(self.seeds) - num_unseen) / len(self.seeds)

    @property
    def requires_value_buffers(self):
        return self.strategy in _VALUE_STRATEGIES

    @property
    def _has_working_seed_buffer(self):
        return not self.sample_full_distribution or self.seed_buffer_size > 0

    @property
    def is_warm(self):
        return self._proportion_filled >= self.rho

    # ---- score updates (reference :185-273) ---------------------------------------------------
    def update_seed_score(self, actor_index, seed, score, max_score, num_steps):
        if self.sample_full_distribution and seed in self.staging_seed_set:
            return self._partial_update_seed_score_buffer(actor_index, seed, score, num_steps, done=True)
        return self._partial_update_seed_score(actor_index, seed, score, max_score, num_steps, done=True)

    def _partial_update_seed_score(self, actor_index, seed, score, max_score, num_steps, done=False):
        seed_idx = self.seed2index.get(seed, -1)
        if seed_idx < 0:
            return 0, None
        partial_score = self.partial_seed_scores[actor_index, seed_idx]
        partial_max = self.partial_seed_max_scores[actor_index, seed_idx]
        partial_steps = self.partial_seed_steps[actor_index, seed_idx]
        running = partial_steps + num_steps
        merged = partial_score + (score - partial_score) * num_steps / float(running)
        merged_max = max(partial_max, max_score)
        if done:
            self.partial_seed_scores[actor_index, seed_idx] = 0.0
            self.partial_seed_max_scores[actor_index, seed_idx] = float("-inf")
            self.partial_seed_steps[actor_index, seed_idx] = 0
            self._state_version = getattr(self, "_state_version", 0) + 1
            self.unseen_seed_weights[seed_idx] = 0.0
            old = self.seed_scores[seed_idx]
            total = self.max_score_coef * merged_max + (1 - self.max_score_coef) * merged
            self.seed_scores[seed_idx] = (1 - self.alpha) * old + self.alpha * total
        else:
            self.partial_seed_scores[actor_index, seed_idx] = merged
            self.partial_seed_max_scores[actor_index, seed_idx] = merged_max
            self.partial_seed_steps[actor_index, seed_idx] = running
            self._partial_dirty.add((actor_index, seed_idx))
        return merged, seed_idx

    @property
    def _next_buffer_index(self):
        if self._proportion_filled < 1.0:
            return self.working_seed_buffer_size
        if self.seed_buffer_priority == "replay_support":
            # Inside update_with_records the weakest slot is memoised on a state version: a candidate level that is
            # rejected (lower score than that slot) leaves scores / staleness / unseen flags untouched, so the next
            # candidate sees the same weights — at 4096 envs an exploratory update tries 4096 candidates, and each
            # sample_weights() is an argsort of the whole buffer.
            memo = getattr(self, "_weakest_memo", None)
            if memo is not None and memo[0] == self._state_version:
                return memo[1]
            idx = self.sample_weights().argmin()
            if memo is not None:
                self._weakest_memo = (self._state_version, idx)
            return idx
        return self.seed_scores.argmin()

    def _partial_update_seed_score_buffer(self, actor_index, seed, score, num_steps, done=False):
        seed_idx = -1
        self.seed2actor[seed].add(actor_index)
        partial_score = self.partial_seed_scores_buffer[actor_index].get(seed, 0)
        partial_steps = self.partial_seed_steps_buffer[actor_index].get(seed, 0)
        running = partial_steps + num_steps
        merged = partial_score + (score - partial_score) * num_steps / float(running)
        if done:
            seed_idx = self._next_buffer_index
            if self.seed_scores[seed_idx] <= merged or self.unseen_seed_weights[seed_idx] > 0:
                self._state_version = getattr(self, "_state_version", 0) + 1
                self.unseen_seed_weights[seed_idx] = 0.0
                self.working_seed_set.discard(self.seeds[seed_idx])
                self.working_seed_set.add(seed)
                self.seeds[seed_idx] = seed
                self.seed2index[seed] = seed_idx
                self.seed_scores[seed_idx] = merged
                self.partial_seed_scores[:, seed_idx] = 0.0
                self.partial_seed_steps[:, seed_idx] = 0
                self.seed_staleness[seed_idx] = self.running_sample_count - self.seed2timestamp_buffer[seed]
                self.working_seed_buffer_size = min(self.working_seed_buffer_size + 1, self.seed_buffer_size)
                if self.track_solvable:
                    self.seed_solvable[seed_idx] = self.staging_seed2solvable.get(seed, True)
            else:
                seed_idx = None
            for a in self.seed2actor[seed]:
                self.partial_seed_scores_buffer[a].pop(seed, None)
                self.partial_seed_steps_buffer[a].pop(seed, None)
            del self.seed2timestamp_buffer[seed]
            del self.seed2actor[seed]
            self.staging_seed_set.remove(seed)
            if self.track_solvable:
                del self.staging_seed2solvable[seed]
        else:
            self.partial_seed_scores_buffer[actor_index][seed] = merged
            self.partial_seed_steps_buffer[actor_index][seed] = running
        return merged, seed_idx

    # ---- rollouts -> records -> scores ----------------------------------------------------------
    def update_with_rollouts(self, rollouts):
        """GPU segmentation + scoring of `rollouts` (a RolloutStorage on a HIP device), then the
        sequential host update (reference :149-183, :486-578)."""
        self._settle_if_streaming()
        if self.strategy in ("random", "off") or not self._has_working_seed_buffer:
            return
        if self.strategy not in _GPU_STRATEGY:
            raise ValueError(f"Unsupported strategy, {self.strategy}")
        recs = self.score_rollouts(rollouts)
        values = None
        if self.grounded_values is not None:
            values = _scoring_values(rollouts)[:, :, 0].float().cpu().numpy()
        self.update_with_records(recs, values)

    def score_rollouts(self, rollouts):
        """Launches the episode scorer on rollouts' device and reads the records back (synchronises)."""
        from dcd_hip import scans
        T, B = rollouts.action_log_dist.shape[:2]
        if self._scorer is None or (self._scorer.T, self._scorer.B) != (T, B) or self._scorer.device != rollouts.device:
            self._scorer = scans.EpisodeScorer(T, B, rollouts.device)
        self._scorer.launch(rollouts.returns, _scoring_values(rollouts), rollouts.rewards, rollouts.masks,
                            rollouts.cliffhanger_masks, rollouts.level_seeds, rollouts.action_log_dist,
                            _GPU_STRATEGY[self.strategy], self.gamma)
        return self._scorer.records()

    def _grounded_score(self, rec, grounded_value, values, actor, seed, done, positive):
        """_average_grounded_{signed,positive}_value_loss (:351-424) for one record.  `actor` is the
        global actor index (sampler bookkeeping); `values` is indexed by the record's own actor."""
        seed_idx = self.seed2index.get(seed, None)
        if self.sample_full_distribution and seed in self.partial_seed_steps_buffer[actor]:
            partial_steps = self.partial_seed_steps_buffer[actor][seed]
        elif seed_idx is not None:
            partial_steps = self.partial_seed_steps[actor][seed_idx]
        else:
            partial_steps = 0
        new_steps = int(rec["num_steps"])
        total_steps = partial_steps + new_steps
        if not (done and grounded_value is not None):
            return 0, 0
        s = int(rec["start_t"])
        col = int(rec["actor"])
        v = values[s:s + 1, col] if self.use_dense_rewards else values[s:s + new_steps, col]
        adv = np.float32(grounded_value) - v.astype(np.float32)
        if positive:
            adv = np.maximum(adv, np.float32(0))
        mean = float(np.float32(adv.astype(np.float64).mean()))
        return (total_steps / new_steps) * mean, float(adv.max())

    def update_with_records(self, recs, values=None, actor_offset=0):
        """Host half of _update_with_rollouts: consume records in order.  `values` ([T+1,B] float32
        numpy) is required for grounded_* strategies; actor_offset shifts local actor indices of a
        remote rank into the global actor space.  The record fields are unpacked to Python lists once: at
        B = 4096 a rollout yields ~8k records and per-record numpy field access dominated the update."""
        self._settle_if_streaming()
        grounded = self.grounded_values is not None
        positive = self.strategy == "grounded_positive_value_loss"
        recs = np.asarray(recs)
        actors, seeds, flags_l = recs["actor"].tolist(), recs["seed"].tolist(), recs["flags"].tolist()
        steps_l, means, maxes = recs["num_steps"].tolist(), recs["mean_score"].tolist(), recs["max_score"].tolist()
        rsums = recs["reward_sum"].tolist() if grounded else None
        staging = self.staging_seed_set if self.sample_full_distribution else ()
        self._state_version = getattr(self, "_state_version", 0) + 1
        self._weakest_memo = (-1, None)                              # memoisation on for the duration of this loop only:
        try:                                                         # outside it callers may edit the arrays directly
            self._consume(actors, seeds, flags_l, steps_l, means, maxes, rsums, recs, values, actor_offset, grounded, positive, staging)
        finally:
            self._weakest_memo = None

    def _consume(self, actors, seeds, flags_l, steps_l, means, maxes, rsums, recs, values, actor_offset, grounded, positive, staging):
        for k in range(len(actors)):
            actor = actors[k] + actor_offset
            seed, flags, num_steps = seeds[k], flags_l[k], steps_l[k]
            if flags & 1:                                            # full episode ending in a done step
                if flags & 2:                                        # cliffhanger: no score update (:528)
                    continue
                grounded_value = None
                if grounded:
                    seed_idx = self.seed2index.get(seed, None)
                    gv = rsums[k]
                    grounded_value = max(self.grounded_values[seed_idx], gv) if seed_idx is not None else gv
                    score, max_score = self._grounded_score(recs[k], grounded_value, values, actor, seed, True, positive)
                else:
                    score, max_score = means[k], maxes[k]
                _, seed_idx = self.update_seed_score(actor, seed, score, max_score, num_steps)
                if seed_idx is not None and grounded and grounded_value is not None:
                    self.grounded_values[seed_idx] = grounded_value
            else:                                                    # trailing partial episode (:551-578)
                if grounded:
                    score, max_score = self._grounded_score(recs[k], None, values, actor, seed, False, positive)
                else:
                    score, max_score = means[k], maxes[k]
                if seed in staging:
                    self._partial_update_seed_score_buffer(actor, seed, score, num_steps)
                else:
                    self._partial_update_seed_score(actor, seed, score, max_score, num_steps)

    def after_update(self):
        """Flush partial scores (policy changed, logits stale) — reference :580-599.  The reference scans the whole
        [num_actors, N] matrix in row-major order; the same entries are visited here, in the same order, from the
        set of pairs that received a partial update."""
        self._settle_if_streaming()
        if not self._has_working_seed_buffer:
            return
        dirty = getattr(self, "_partial_dirty", None)
        if dirty is None:                                            # sampler restored from an older checkpoint
            pairs = list(zip(*np.nonzero(self.partial_seed_scores)))
        else:
            pairs = sorted(dirty)                                    # row-major == actor-major
        for actor_index, seed_idx in pairs:
            if self.partial_seed_scores[actor_index, seed_idx] != 0:
                self.update_seed_score(int(actor_index), self.seeds[seed_idx], 0, float("-inf"), 0)
        for actor_index, seed_idx in pairs:                          # .fill(0) of the reference, on the touched entries
            self.partial_seed_scores[actor_index, seed_idx] = 0
            self.partial_seed_steps[actor_index, seed_idx] = 0
        self._partial_dirty = set()
        if self.sample_full_distribution:
            for actor_index in range(self.num_actors):
                buf = self.partial_seed_scores_buffer[actor_index]
                if not buf:
                    continue
                for seed in list(buf.keys()):
                    if buf.get(seed, 0) > 0:
                        self.update_seed_score(actor_index, seed, 0, float("-inf"), 0)

    # ---- sampling (reference :601-750) -----------------------------------------------------------
    def _update_staleness(self, selected_idx):
        self._state_version = getattr(self, "_state_version", 0) + 1
        if self.staleness_coef > 0:
            self.seed_staleness = self.seed_staleness + 1
            self.seed_staleness[selected_idx] = 0

    # ---- speculative prefetch of the next update's first draws ------------------------------------------------------------
    # What an update asks of the sampler first — sample_replay_decision(), then, on a replay update, one level per actor
    # (adversarial_runner.py:616-622,463) — depends only on the sampler's state after the previous update's after_update() and
    # on the position of the np.random stream.  `prefetch_next_update(n)` computes both on a background host thread from a
    # SHADOW of that state (copies of the arrays a draw touches, a private RandomState started at the global stream's
    # position) while the GPU runs the PPO update; the sequential draw loop is C code outside the GIL.  The next
    # sample_replay_decision() / sample_replay_levels(n) consume the result ONLY IF nothing has moved in between — same
    # global np.random state, same scores / staleness / seeds — and then install exactly what the sequential calls would have
    # left behind (staleness, versions, the global stream advanced past the same doubles).  Otherwise the speculation is
    # dropped and the calls run as always: bit-identical either way (tests/test_sampler_prefetch.py).
    @property
    def _rng(self):
        return self.__dict__.get("_shadow_rng") or np.random

    _PREFETCH_ARRAYS = ("seed_scores", "seed_staleness", "unseen_seed_weights", "seeds")

    def prefetch_next_update(self, n_initial, stream_extra=None):
        """stream_extra: how many draws BEYOND the first n_initial the thread keeps producing afterwards (the per-episode
        re-sampling of the replay rollout, adversarial_runner.py:551-558, asks for them a few at a time): the draws are one
        sequential stream however the requests cut it, so they are speculated in chunks of STREAM_CHUNK with a staleness
        snapshot per chunk; a request pops its draws and burns the same doubles from the global stream (which doubles as the
        validity check); `_settle` installs the staleness after exactly the draws that were popped."""
        import copy
        import threading
        self._drop_prefetch()
        rng_transforms = self.score_transform == "max" or (self.staleness_coef > 0 and self.staleness_transform == "max")
        if n_initial <= 1 or rng_transforms or not self._has_working_seed_buffer:
            return False
        shadow = copy.copy(self)
        for k in self._PREFETCH_ARRAYS:
            setattr(shadow, k, np.array(getattr(self, k), copy=True))
        shadow.__dict__.pop("_prefetch", None)
        state0 = np.random.get_state()
        shadow._shadow_rng = np.random.RandomState()
        shadow._shadow_rng.set_state(state0)
        job = {"n": int(n_initial), "rng0": state0, "snap": {k: np.array(getattr(self, k), copy=True) for k in self._PREFETCH_ARRAYS},
               "filled": self._proportion_filled, "shadow": shadow, "stage": 0, "initial_done": threading.Event(),
               "cv": threading.Condition(), "finished": False, "stop": False, "extra": int(stream_extra or 0),
               "q_seeds": [], "q_uniforms": [], "snaps": [], "produced": 0, "used": 0}

        def work():
            try:
                job["decision"] = shadow._sample_replay_decision_now()
                job["rng1"] = shadow._shadow_rng.get_state()
                if job["decision"]:
                    job["seeds_out"] = shadow._sample_replay_levels_now(job["n"], True)
                    job["rng2"] = shadow._shadow_rng.get_state()
                    # the consumer installs THIS copy: once the event is set the thread goes on mutating / rebinding
                    # shadow.seed_staleness with streamed draws, so the shadow itself is never read from the other side
                    job["stale_after_initial"] = shadow.seed_staleness.copy()
                    job["initial_done"].set()
                    self._stream_ahead(job, shadow)
            except BaseException as exc:                          # noqa: BLE001  (a failed speculation is simply not used)
                job["error"] = exc
            finally:
                job["initial_done"].set()
                with job["cv"]:
                    job["finished"] = True
                    job["cv"].notify_all()
        job["thread"] = threading.Thread(target=work, daemon=True)
        job["thread"].start()
        self._prefetch = job
        return True

    STREAM_CHUNK = 512

    def _stream_ahead(self, job, shadow):
        """Thread side: draws n_initial+1, n_initial+2, ... on the shadow, in chunks, until `extra` are out or the stream is stopped."""
        rec = _RecordingRng(shadow._shadow_rng)
        shadow._shadow_rng = rec
        job["snaps"].append(job["stale_after_initial"])            # staleness after the initial draws = before chunk 0
        while job["produced"] < job["extra"] and not job["stop"]:
            c = min(self.STREAM_CHUNK, job["extra"] - job["produced"])
            rec.doubles, rec.unsafe = [], False
            seeds = shadow._sample_replay_levels_now(c, True)
            if rec.unsafe or sum(len(d) for d in rec.doubles) != c:   # a draw went through numpy's own choice(): not replayable
                break
            with job["cv"]:
                job["q_seeds"].extend(seeds)
                job["q_uniforms"].append(np.concatenate(rec.doubles))
                job["snaps"].append(shadow.seed_staleness.copy())
                job["produced"] += c
                job["cv"].notify_all()

    def _stream_pop(self, job, n):
        """Consumer side: the next n draws of the stream, or None when it cannot serve them (the caller then settles and draws
        in program order).  The global np.random stream is advanced by the n doubles these draws consumed — and must hold
        exactly those doubles, else something overtook the speculation."""
        if n == 0:
            return []
        with job["cv"]:
            while job["produced"] < job["used"] + n and not job["finished"]:
                job["cv"].wait()
            if job["produced"] < job["used"] + n:
                return None
            lo = job["used"]
            seeds = job["q_seeds"][lo:lo + n]
        state = np.random.get_state()
        burned = np.random.random_sample(n)
        have = self._stream_uniforms(job, lo, n)
        if not np.array_equal(burned, have):
            np.random.set_state(state)                            # untouched: the sequential path draws these doubles itself
            return None
        job["used"] = lo + n
        return seeds

    def peek_stream(self, min_k, max_k):
        """The next draws of the OPEN stream (the ones `sample_replay_levels` would hand out next) without consuming them:
        waits until at least `min_k` are speculated, returns up to `max_k` of what is there — or None when no stream is open
        or it ends before `min_k`.  The device-side replay queue (adversarial_runner.py: _queued_replay_rollout) uploads them,
        lets the rollout consume a prefix on the GPU and then pops exactly that prefix with sample_replay_levels(consumed)."""
        job = self.__dict__.get("_prefetch")
        if job is None or job["stage"] != 2:
            return None
        with job["cv"]:
            while job["produced"] < job["used"] + min_k and not job["finished"]:
                job["cv"].wait()
            have = job["produced"] - job["used"]
            if have < min_k:
                return None
            return list(job["q_seeds"][job["used"]:job["used"] + min(have, max_k)])

    def _settle_if_streaming(self):
        """State-changing entry points end an OPEN stream first (its staleness is installed on settling); a speculation
        that has not been consumed yet (stage 0 / 1) is left to its own validity check."""
        job = self.__dict__.get("_prefetch")
        if job is not None and job["stage"] == 2:
            self._settle()

    def _stream_uniforms(self, job, lo, n):
        """Doubles lo .. lo+n of the stream (chunk k holds doubles [k * STREAM_CHUNK, (k + 1) * STREAM_CHUNK))."""
        C, parts = self.STREAM_CHUNK, []
        while n > 0:
            k, o = divmod(lo, C)
            take = min(n, C - o)
            parts.append(job["q_uniforms"][k][o:o + take])
            lo, n = lo + take, n - take
        return parts[0] if len(parts) == 1 else np.concatenate(parts)

    def _settle(self):
        """Ends any speculation: joins the thread and, if a stream served per-step draws, installs the staleness after exactly
        those draws (snapshot at the chunk boundary below + a replay of the remainder with the recorded doubles)."""
        job = self.__dict__.pop("_prefetch", None)
        if job is None:
            return
        job["stop"] = True
        job["thread"].join()
        used = job["used"]
        if job["stage"] == 2 and used:
            C = self.STREAM_CHUNK
            a, r = divmod(used, C)
            tmp = job["shadow"]
            tmp.seed_staleness = job["snaps"][a].copy()
            if r:
                uni = self._stream_uniforms(job, a * C, r)
                tmp._shadow_rng = None
                again = tmp._sample_replay_levels_now(r, True, uniforms=uni)
                if again != job["q_seeds"][a * C:a * C + r]:
                    raise RuntimeError("LevelSampler: the replay of the speculated draws diverged from the draws that were handed out")
            self.seed_staleness = tmp.seed_staleness
            self._state_version = getattr(self, "_state_version", 0) + 1

    def _drop_prefetch(self):
        self._settle()

    @staticmethod
    def _same_rng(a, b):
        return a[0] == b[0] and a[2:] == b[2:] and np.array_equal(a[1], b[1])

    def _prefetch_valid(self, job, rng_key):
        """The speculation started from exactly the state the sequential call sees now."""
        job["initial_done"].wait()
        if "error" in job or not self._same_rng(np.random.get_state(), job[rng_key]) or self._proportion_filled != job["filled"]:
            return False
        return all(np.array_equal(getattr(self, k), job["snap"][k]) for k in self._PREFETCH_ARRAYS)

    def sample_replay_decision(self):
        job = self.__dict__.get("_prefetch")
        if job is not None and job["stage"] == 0:
            if self._prefetch_valid(job, "rng0"):
                np.random.set_state(job["rng1"])                  # the stream past the (at most one) double the decision drew
                job["stage"] = 1
                if not job["decision"]:
                    self._settle()
                self.prefetch_hits = getattr(self, "prefetch_hits", 0) + 1
                return job["decision"]
            self._drop_prefetch()
        elif job is not None:
            self._drop_prefetch()
        return self._sample_replay_decision_now()

    def _sample_replay_decision_now(self):
        if self.sample_full_distribution:
            filled = self._proportion_filled
            if self.seed_buffer_size <= 0:
                return False
            if self.replay_schedule == "fixed":
                return bool(filled >= self.rho and self._rng.rand() < self.replay_prob)
            return bool(filled >= self.rho and self._rng.rand() < min(filled, self.replay_prob))
        if self.replay_schedule == "fixed":
            seen = self._proportion_filled
            if seen >= self.rho:
                if self._rng.rand() < self.replay_prob or not seen < 1.0:
                    return True
            return False
        seen = self._proportion_filled
        return bool(seen >= self.rho and self._rng.rand() < seen)

    def observe_external_unseen_sample(self, seeds, solvable=None):
        self._settle_if_streaming()
        for i, seed in enumerate(seeds):
            self.running_sample_count += 1
            if not (seed in self.staging_seed_set or seed in self.working_seed_set):
                self.seed2timestamp_buffer[seed] = self.running_sample_count
                self.staging_seed_set.add(seed)
                if solvable is not None:
                    if not self.track_solvable:
                        self._init_solvable_tracking()
                    self.staging_seed2solvable[seed] = solvable[i]
            else:
                seed_idx = self.seed2index.get(seed, None)
                if seed_idx is not None:
                    self._update_staleness(seed_idx)

    def sample_replay_level(self, update_staleness=True):
        self._settle_if_streaming()
        return self._sample_replay_level(update_staleness=update_staleness)

    def _replay_distribution(self):
        w = self.sample_weights()
        if np.isclose(np.sum(w), 0):
            w = np.ones_like(self.seeds, dtype=np.float64) / len(self.seeds)
            w = w * (1 - self.unseen_seed_weights)
            w /= np.sum(w)
        elif np.sum(w, 0) != 1.0:
            w = w / np.sum(w, 0)
        return w

    def _sample_replay_level(self, update_staleness=True):
        w = self._replay_distribution()
        seed_idx = self._rng.choice(range(len(self.seeds)), 1, p=w)[0]
        seed = self.seeds[seed_idx]
        if update_staleness:
            self._update_staleness(seed_idx)
        return int(seed)

    def sample_replay_levels(self, n, update_staleness=True):
        """`_sample_replay_levels_now`, or its prefetched result when one is pending and still valid (see above)."""
        job = self.__dict__.get("_prefetch")
        if job is not None and job["stage"] == 2:                 # per-step draws of the rollout: the next ones of the stream
            got = self._stream_pop(job, n) if update_staleness else None
            if got is not None:
                self.stream_hits = getattr(self, "stream_hits", 0) + (1 if n else 0)
                return got
            self._settle()
        elif job is not None:
            job["initial_done"].wait()
            ok = (job["stage"] == 1 and job.get("decision") and n == job["n"] and update_staleness and "error" not in job
                  and self._same_rng(np.random.get_state(), job["rng1"])
                  and all(np.array_equal(getattr(self, k), job["snap"][k]) for k in self._PREFETCH_ARRAYS))
            if ok:
                np.random.set_state(job["rng2"])
                self.seed_staleness = job["stale_after_initial"].copy()
                self._state_version = getattr(self, "_state_version", 0) + 1
                self.prefetch_hits = getattr(self, "prefetch_hits", 0) + 1
                job["stage"] = 2
                if not job["extra"]:
                    self._settle()
                return job["seeds_out"]
            self._settle()
        return self._sample_replay_levels_now(n, update_staleness)

    def _sample_replay_levels_now(self, n, update_staleness=True, uniforms=None):
        """n sequential draws of sample_replay_level() (the runner's per-env loop, adversarial_runner.py:463) — same
        draws, same RNG consumption, same staleness bookkeeping as n separate calls, without recomputing what cannot
        change between them: seed scores are fixed during a batch of draws, so the score half of `sample_weights`
        (for `rank`: an argsort of the whole buffer) is formed once; only the staleness half, the mix, the
        normalisation of `_replay_distribution` and the inverse-CDF draw (what `np.random.choice(range(N), 1, p=w)`
        does: cumsum, divide by the last entry, one `random_sample`, `searchsorted(side="right")`) run per draw.
        326 -> ~40 us per draw at N = 4000: a replay rollout of 4096 envs needs 4096 draws up front."""
        rng_transforms = self.score_transform == "max" or (self.staleness_coef > 0 and self.staleness_transform == "max")
        if (n <= 1 and uniforms is None) or rng_transforms:         # 'max' consumes the RNG inside the transform
            return [self._sample_replay_level(update_staleness=update_staleness) for _ in range(n)]
        if n == 0:
            return []
        # the score half of the weights depends on seed_scores / unseen_seed_weights only, which no draw changes: a replay
        # rollout calls this once per step (256 argsorts of the buffer), so it is kept while those two arrays are unchanged
        cached = getattr(self, "_score_half", None)
        if (cached is not None and cached[0] == (self.score_transform, self.temperature)
                and np.array_equal(cached[1], self.seed_scores) and np.array_equal(cached[2], self.unseen_seed_weights)):
            unseen_keep, ws = cached[3], cached[4].copy()
        else:
            unseen_keep = 1 - self.unseen_seed_weights
            ws = self._score_transform(self.score_transform, self.temperature, self.seed_scores)
            ws = ws * unseen_keep
            z = np.sum(ws)
            if z > 0:
                ws /= z
            else:
                ws = np.ones_like(ws, dtype=np.float64) / len(ws)
                ws = ws * unseen_keep
                ws /= np.sum(ws)
            self._score_half = ((self.score_transform, self.temperature), np.array(self.seed_scores, copy=True),
                                np.array(self.unseen_seed_weights, copy=True), unseen_keep, ws.copy())
        c, N = self.staleness_coef, len(self.seeds)
        w_scores = (1 - c) * ws if c > 0 else ws                     # constant across the batch of draws
        # 'power' with temperature 1 and eps 0 (staleness_coef > 0): (clip(s, 0) + 0) ** 1.0 is s itself, bit for bit
        plain_staleness = self.staleness_transform == "power" and self.staleness_temperature == 1.0
        if uniforms is None:
            uniforms = self._rng.random_sample(n)                    # the n doubles n calls of choice(..., 1, p=w) would draw
        out, k = [], 0
        if c > 0 and plain_staleness and _native_draws_available():
            # the same loop in C (dcd_sampler_draws, csrc/host_utils.cpp): same IEEE operations in the same order
            import ctypes as C
            from dcd_hip import lib
            st = np.ascontiguousarray(self.seed_staleness, dtype=np.float64).copy()
            idx = np.empty(n, dtype=np.int64)
            p_ = lambda a: a.ctypes.data_as(C.c_void_p)
            wsc, keep = np.ascontiguousarray(w_scores, dtype=np.float64), np.ascontiguousarray(unseen_keep, dtype=np.float64)
            k = int(lib().dcd_sampler_draws(C.c_int64(N), C.c_int64(n), p_(wsc), C.c_double(c), p_(st), p_(keep), p_(uniforms),
                                            C.c_int32(int(update_staleness)), p_(idx)))
            self.seed_staleness = st
            if update_staleness and k:
                self._state_version = getattr(self, "_state_version", 0) + 1
            out = [int(self.seeds[i]) for i in idx[:k]]
        for j in range(k, n):                                        # numpy statement (and continuation of the C loop)
            if c > 0:
                if plain_staleness and self.seed_staleness.min() >= 0:
                    sw = self.seed_staleness * unseen_keep
                else:
                    sw = self._score_transform(self.staleness_transform, self.staleness_temperature, self.seed_staleness)
                    sw = sw * unseen_keep
                z = np.sum(sw)
                if z > 0:
                    sw /= z
                else:
                    sw = 1.0 / len(sw) * unseen_keep
                w = w_scores + c * sw
            else:
                w = ws
            total = np.sum(w)
            if abs(total) <= 1e-8:                                   # np.isclose(total, 0) for a scalar
                w = np.ones_like(self.seeds, dtype=np.float64) / N
                w = w * unseen_keep
                w /= np.sum(w)
            elif total != 1.0:
                w = w / total
            cdf = w.cumsum()
            if not (abs(cdf[-1] - 1.0) <= 1e-8) or w.min() < 0:     # let numpy raise exactly as the sequential path would
                seed_idx = self._rng.choice(range(N), 1, p=w)[0]
            else:
                cdf /= cdf[-1]
                seed_idx = int(cdf.searchsorted(uniforms[j:j + 1], side="right")[0])
            if update_staleness:
                self._update_staleness(seed_idx)
            out.append(int(self.seeds[seed_idx]))
        return out

    def _sample_unseen_level(self):
        self._settle_if_streaming()
        if self.sample_full_distribution:
            seed = int(np.random.randint(1, INT32_MAX))
            while seed in self.staging_seed_set or seed in self.working_seed_set:
                seed = int(np.random.randint(1, INT32_MAX))
            self.seed2timestamp_buffer[seed] = self.running_sample_count
            self.staging_seed_set.add(seed)
        else:
            w = self.unseen_seed_weights / self.unseen_seed_weights.sum()
            seed_idx = np.random.choice(range(len(self.seeds)), 1, p=w)[0]
            seed = self.seeds[seed_idx]
            self._update_staleness(seed_idx)
        return int(seed)

    def sample(self, strategy=None):
        if strategy == "full_distribution":
            raise ValueError("One-off sampling via full_distribution strategy is not supported.")
        self._settle_if_streaming()
        self.running_sample_count += 1
        strategy = strategy or self.strategy
        if not self.sample_full_distribution:
            if strategy == "random":
                return int(self.seeds[np.random.choice(range(len(self.seeds)))])
            if strategy == "sequential":
                seed_idx = self.next_seed_index
                self.next_seed_index = (self.next_seed_index + 1) % len(self.seeds)
                return int(self.seeds[seed_idx])
        return self._sample_replay_level() if self.sample_replay_decision() else self._sample_unseen_level()

    def sample_weights(self):
        self._settle_if_streaming()
        weights = self._score_transform(self.score_transform, self.temperature, self.seed_scores)
        weights = weights * (1 - self.unseen_seed_weights)
        z = np.sum(weights)
        if z > 0:
            weights /= z
        else:
            weights = np.ones_like(weights, dtype=np.float64) / len(weights)
            weights = weights * (1 - self.unseen_seed_weights)
            weights /= np.sum(weights)
        if self.staleness_coef > 0:
            sw = self._score_transform(self.staleness_transform, self.staleness_temperature, self.seed_staleness)
            sw = sw * (1 - self.unseen_seed_weights)
            z = np.sum(sw)
            if z > 0:
                sw /= z
            else:
                sw = 1.0 / len(sw) * (1 - self.unseen_seed_weights)
            weights = (1 - self.staleness_coef) * weights + self.staleness_coef * sw
        return weights

    def _score_transform(self, transform, temperature, scores):
        if transform == "constant":
            weights = np.ones_like(scores)
        if transform == "max":
            weights = np.zeros_like(scores)
            scores = scores[:]
            scores[self.unseen_seed_weights > 0] = -float("inf")
            argmax = np.random.choice(np.flatnonzero(np.isclose(scores, scores.max())))
            weights[argmax] = 1.0
        elif transform == "eps_greedy":
            weights = np.zeros_like(scores)
            weights[scores.argmax()] = 1.0 - self.eps
            weights += self.eps / len(self.seeds)
        elif transform in ("rank", "match_rank"):
            base = scores if transform == "rank" else np.array([(1 - s) * s for s in scores])
            temp = np.flip(base.argsort())
            # weights = 1 / ranks ** (1 / temperature) with ranks[temp] = 1..N (:765-769): the value for rank k is the
            # same elementwise expression whatever the order, so it is tabulated once per (N, temperature) and scattered
            key = (len(temp), temperature)
            tab = self._rank_weight_tables.get(key) if hasattr(self, "_rank_weight_tables") else None
            if tab is None:
                tab = 1 / (np.arange(len(temp)) + 1) ** (1.0 / temperature)
                if not hasattr(self, "_rank_weight_tables"):
                    self._rank_weight_tables = {}
                self._rank_weight_tables[key] = tab
            weights = np.empty(len(temp), dtype=tab.dtype)
            weights[temp] = tab
        elif transform == "power":
            eps = 0 if self.staleness_coef > 0 else 1e-3
            weights = (np.array(scores).clip(0) + eps) ** (1.0 / temperature)
        elif transform == "softmax":
            weights = np.exp(np.array(scores) / temperature)
        elif transform == "match":
            weights = np.array([(1 - s) * s for s in scores]) ** (1.0 / temperature)
        return weights

    @property
    def solvable_mass(self):
        if self.track_solvable:
            return np.sum(self.sample_weights()[self.seed_solvable])
        return 1.0

    @property
    def max_score(self):
        return max(self.seed_scores)

    def __getstate__(self):                    # checkpoints pickle the sampler (adversarial_runner.py:214)
        self._settle_if_streaming()            # an open stream's staleness is installed before the state is copied
        d = dict(self.__dict__)
        d["_scorer"] = None
        d.pop("_prefetch", None)
        d.pop("_shadow_rng", None)
        return d
