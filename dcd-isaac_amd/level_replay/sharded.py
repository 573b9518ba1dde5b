"""Level replay for an env batch sharded over ranks (BASELINE config 3: PLR-robust, 4096 envs
per GPU on 8 GPUs).  The reference has one process and one `LevelSampler`; here every rank owns B envs and

  * ONE `LevelSampler` lives on rank 0, sized for world x B actors (global actor = rank * B + local actor);
  * every rank scores its own rollout on its own GPU (`dcd_episode_scores`), the per-episode records are
    all-gathered and rank 0 consumes them rank-major == the order one process with world x B envs would
    (`level_sampler.py:497-578` iterates actors in order);
  * sampling decisions (replay or not, the replay seeds of every env, per-step re-sampling of finished envs)
    are drawn on rank 0 in global env order and broadcast;
  * the `LevelStore` is REPLICATED: inserts are all-gathered and applied in global env order on every rank,
    so seed numbering and parent lineage are identical everywhere and `get_level` is local (a level is 675 B).

Both classes keep the method set `AdversarialRunner` uses, so the runner code path is the same for one
process and for N.  Every method that communicates is a COLLECTIVE: all ranks call it, in the same order —
which the runner guarantees because its control flow depends only on broadcast decisions.
"""
import numpy as np
import torch
import torch.distributed as dist

from .level_sampler import LevelSampler, _GPU_STRATEGY, _scoring_values
from .level_store import LevelStore


def _world(group):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def _rank(group):
    return dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0


def _all_gather_object(obj, group):
    out = [None] * _world(group)
    dist.all_gather_object(out, obj, group=group)
    return out


def _broadcast_object(obj, group):
    box = [obj]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return box[0]


class ShardedLevelStore(LevelStore):
    """Replicated store: `insert` takes this rank's levels, returns this rank's seeds."""

    def __init__(self, max_size=None, data_info={}, group=None):
        super().__init__(max_size=max_size, data_info=data_info)
        self.group = group

    def insert(self, level, parent_seeds=None):
        if not (hasattr(level, "__iter__") and not isinstance(level, (str, bytes))):
            raise TypeError("ShardedLevelStore.insert takes this rank's list of levels")
        level = list(level)
        parts = _all_gather_object((level, None if parent_seeds is None else [int(p) for p in parent_seeds]), self.group)
        mine, r = None, _rank(self.group)
        for k, (lv, ps) in enumerate(parts):                               # global env order
            seeds = LevelStore.insert(self, lv, parent_seeds=ps)
            if k == r:
                mine = seeds
        return mine

    def __getstate__(self):
        d = dict(self.__dict__)
        d["group"] = None
        return d


class ShardedLevelSampler:
    """The runner-facing proxy of rank 0's LevelSampler.  `num_actors` in `plr_args` is the LOCAL env count."""

    def __init__(self, plr_args, group=None):
        self.group = group
        self.world, self.rank = _world(group), _rank(group)
        self.local_actors = int(plr_args["num_actors"])
        self.strategy = plr_args.get("strategy", "random")
        self.gamma = plr_args.get("gamma", 0.999)
        self.sample_full_distribution = plr_args.get("sample_full_distribution", False)
        self.sampler = None
        if self.rank == 0:
            kw = dict(plr_args)
            kw["num_actors"] = self.local_actors * self.world
            self.sampler = LevelSampler(**kw)
        self._scorer = None

    # ---- rank 0 -> all ---------------------------------------------------------------------------
    def _from_root(self, fn):
        return _broadcast_object(fn(self.sampler) if self.rank == 0 else None, self.group)

    def sample_replay_decision(self):
        from util import distributed as D
        d = int(bool(self.sampler.sample_replay_decision())) if self.rank == 0 else 0
        return bool(D.broadcast_ints([d], 1, group=self.group)[0])

    def prefetch_next_update(self, n_local, stream_extra=None):
        """LevelSampler.prefetch_next_update on rank 0 for the whole job's first draws (world x local actors, the count
        `sample_replay_levels` will sum up) and its share of per-step draws; no collective: the other ranks do nothing."""
        if self.rank == 0 and n_local == self.local_actors:
            return self.sampler.prefetch_next_update(self.local_actors * self.world, stream_extra=(stream_extra or 0) * self.world)
        return False

    def sample_replay_levels(self, n, update_staleness=True):
        """n replay seeds for THIS rank; rank 0 draws sum(n) sequentially in global env order (the draws are sequential
        by definition: each one ages the whole buffer, level_sampler.py:664-680).  Three small tensor collectives
        (counts, seeds) — on the GPU path RCCL, no pickling."""
        from util import distributed as D
        counts = [c[0] for c in D.gather_int_lists([int(n)], group=self.group)]
        total = sum(counts)
        seeds = self.sampler.sample_replay_levels(total, update_staleness=update_staleness) if (self.rank == 0 and total) else []
        seeds = D.broadcast_ints(seeds, total, group=self.group) if total else []
        off = sum(counts[:self.rank])
        return seeds[off:off + n]

    def sample_replay_level(self, update_staleness=True):
        return self.sample_replay_levels(1, update_staleness=update_staleness)[0]

    def sample_weights(self):
        return self._from_root(lambda s: s.sample_weights())

    @property
    def seeds(self):
        return self._from_root(lambda s: s.seeds)

    @property
    def seed_buffer_size(self):
        return self._from_root(lambda s: s.seed_buffer_size)

    @property
    def solvable_mass(self):
        return self._from_root(lambda s: float(s.solvable_mass))

    @property
    def max_score(self):
        return self._from_root(lambda s: float(s.max_score))

    @property
    def is_warm(self):
        return self._from_root(lambda s: bool(s.is_warm))

    @property
    def requires_value_buffers(self):
        return self.strategy in ("gae", "value_l1", "signed_value_loss", "positive_value_loss",
                                 "grounded_signed_value_loss", "grounded_positive_value_loss", "one_step_td_error")

    # ---- all -> rank 0 ---------------------------------------------------------------------------
    def observe_external_unseen_sample(self, seeds, solvable=None):
        from util import distributed as D
        # seed and solvable flag of every new level in one ragged int gather: 2*seed + solvable (-1: no flags given)
        code = [2 * int(s) + (int(bool(x)) if solvable is not None else 0) for s, x in
                zip(seeds, solvable if solvable is not None else [0] * len(seeds))]
        parts = D.gather_int_lists([1 if solvable is not None else 0] + code, group=self.group)
        if self.rank == 0:
            all_seeds = [c >> 1 for p in parts for c in p[1:]]
            all_solv = [bool(c & 1) for p in parts for c in p[1:]] if parts[0][0] else None
            self.sampler.observe_external_unseen_sample(all_seeds, all_solv)

    def score_rollouts(self, rollouts, to_host=True):
        from dcd_hip import scans
        T, B = rollouts.action_log_dist.shape[:2]
        if self._scorer is None or (self._scorer.T, self._scorer.B) != (T, B) or self._scorer.device != rollouts.device:
            self._scorer = scans.EpisodeScorer(T, B, rollouts.device)
        self._scorer.launch(rollouts.returns, _scoring_values(rollouts), rollouts.rewards, rollouts.masks,
                            rollouts.cliffhanger_masks, rollouts.level_seeds, rollouts.action_log_dist,
                            _GPU_STRATEGY[self.strategy], self.gamma)
        return self._scorer.records() if to_host else None

    def update_with_rollouts(self, rollouts):
        if self.strategy in ("random", "off"):
            return
        if self.strategy not in _GPU_STRATEGY:
            raise ValueError(f"Unsupported strategy, {self.strategy}")
        from util import distributed as D
        self.score_rollouts(rollouts, to_host=False)
        values = None
        if self.strategy.startswith("grounded"):
            values = _scoring_values(rollouts)[:, :, 0].float().cpu().numpy()
        # the records leave the scorer's device buffers in ONE fixed-capacity all-gather; only rank 0 reads them back
        sc = self._scorer
        counts, recv = D.gather_records_device(sc.recs, sc.n_recs, group=self.group, cap=4 * sc.B)
        vals = _all_gather_object(values, self.group) if values is not None else [None] * self.world
        if self.rank == 0:
            from dcd_hip.scans import EPISODE_REC
            body = recv[:, 1:].cpu().numpy()
            for r, c in enumerate(counts):
                part = body[r, :c * 8].copy().view(EPISODE_REC).reshape(c)
                self.sampler.update_with_records(part, vals[r], actor_offset=r * self.local_actors)

    def update_with_records(self, recs, values=None, device="cpu"):
        """This rank's records (local actor indices) -> rank 0's sampler, consumed rank-major.  The records
        travel as ONE padded int32 all-gather on `device` (RCCL over xGMI on the GPU path); the value
        predictions (grounded strategies only) ride along as host objects."""
        from util import distributed as D
        parts = D.all_gather_records(np.ascontiguousarray(recs), device, group=self.group)
        vals = _all_gather_object(values, self.group) if values is not None else [None] * self.world
        if self.rank == 0:
            for r, part in enumerate(parts):
                self.sampler.update_with_records(part, vals[r], actor_offset=r * self.local_actors)

    def after_update(self):
        if self.rank == 0:
            self.sampler.after_update()

    def __getstate__(self):                    # checkpoints: rank 0 saves the real sampler state
        d = dict(self.__dict__)
        d["group"], d["_scorer"] = None, None
        return d
