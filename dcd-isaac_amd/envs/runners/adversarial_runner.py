"""AdversarialRunner — the rollout-and-regret loop (reference: envs/runners/adversarial_runner.py:27-896)
with the same constructor, `run()` call order, stats keys and `state_dict()` layout, driven through
the device-level API so that a rollout performs NO host synchronisation per env step:

  reference (per step, :497-588)                       here
  ------------------------------------------------     ------------------------------------------------
  agent.act on GPU, action.cpu()  (D2H sync)           agent.act on GPU, action stays on device
  venv.step_env over pipes to N worker processes       dcd_mg_step: one kernel for all B envs
  Python loops over infos (episode returns, masks)     kernel writes masks / episode stats to the storage
  agent.insert: ~12 H2D copies                         kernel writes obs/reward straight into storage[step]
  _get_rollout_return_stats: Python loop over B        dcd_rollout_return_stats
  _compute_env_return (regret) on CPU tensors          dcd_env_return

Host decisions that the reference takes inside the step loop (PLR per-episode level re-sampling,
:551-558) still need the done mask on the host and therefore synchronise once per step in replay
rollouts; PAIRED / DR / non-replay rollouts are sync-free until the rollout ends.
"""
import os
from collections import defaultdict, deque

import numpy as np
import torch

import dcd_hip
from dcd_hip import scans
from level_replay import LevelSampler, LevelStore
from util import is_discrete_actions


class RunningMeanStd:
    """Parallel-variance running moments (what baselines.common.running_mean_std provides to
    adversarial_runner.py:84,666-668)."""

    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, "float64")
        self.var = np.ones(shape, "float64")
        self.count = epsilon

    def update(self, x):
        x = np.asarray(x, dtype=np.float64)
        b_mean, b_var, b_count = x.mean(axis=0), x.var(axis=0), x.shape[0]
        delta = b_mean - self.mean
        tot = self.count + b_count
        m2 = self.var * self.count + b_var * b_count + np.square(delta) * self.count * b_count / tot
        self.mean, self.var, self.count = self.mean + delta * b_count / tot, m2 / tot, tot


def _startup_env(name):
    """The value `name` had when the process STARTED (/proc/self/environ: the initial environment block, which later
    os.environ / setenv writes do not touch).  hipBLASLt reads TENSILE_STREAMK_DATA_PARALLEL once per process, so a value set
    from Python after the first GEMM would pass an os.environ check and still deadlock the GPU."""
    try:
        with open("/proc/self/environ", "rb") as f:
            for item in f.read().split(b"\0"):
                k, _, v = item.partition(b"=")
                if k == name.encode():
                    return v.decode()
        return None
    except OSError:
        return os.environ.get(name)


class AdversarialRunner:
    def __init__(self, args, venv, agent, ued_venv=None, adversary_agent=None, adversary_env=None,
                 flexible_protagonist=False, train=False, plr_args=None, device="cpu", process_group=None):
        """`process_group` (not in the reference): the env batch is sharded over its ranks, `venv` holds this
        rank's `args.num_processes` envs; level replay then goes through level_replay.sharded (one sampler on
        rank 0, replicated level store) and host-side random decisions are taken on rank 0."""
        self.args = args
        self.process_group = process_group
        self.world_size = 1
        # a group of ONE rank takes the same code path as a group of eight — every collective runs (as the identity): that is how
        # the RCCL path is exercised on a one-GPU box (tests/test_gpu_rccl_single.py)
        self.distributed = process_group is not None
        self.prefetch_draws = os.environ.get("DCD_PREFETCH_DRAWS", "1") != "0"
        # DCD_REPLAY_QUEUE=1 (default; one process only): a replay rollout takes the levels of its finished episodes from a
        # device-resident queue of the sampler's speculated draws — ONE graph and one synchronisation per rollout instead of one
        # per step (_queued_replay_rollout); the stream is then speculated 8 draws per env ahead instead of 3
        self.replay_queue = os.environ.get("DCD_REPLAY_QUEUE", "1") != "0" and not self.distributed
        self.prefetch_stream_factor = int(os.environ.get("DCD_PREFETCH_STREAM", "8" if self.replay_queue else "3"))
        if process_group is not None:
            import torch.distributed as dist
            self.world_size = dist.get_world_size(process_group)
        self.venv = venv
        self.ued_venv = venv if ued_venv is None else ued_venv
        self._pending_updates, self._update_streams, self._deferred_host = [], [], []
        # DCD_CONCURRENT_PPO=1 (opt-in): the PPO updates of one run() on three streams (_flush_updates).  Refused unless the
        # process was started with TENSILE_STREAMK_DATA_PARALLEL=1: GEMMs of several streams deadlock otherwise (see there).
        self._concurrent_updates = not self.distributed and os.environ.get("DCD_CONCURRENT_PPO", "0") == "1"
        if self._concurrent_updates and _startup_env("TENSILE_STREAMK_DATA_PARALLEL") != "1":
            raise ValueError("DCD_CONCURRENT_PPO=1 needs TENSILE_STREAMK_DATA_PARALLEL=1 in the environment the process STARTS with: "
                             "hipBLASLt's StreamK GEMMs spin on flags between workgroups and deadlock when GEMMs of several streams "
                             "are in flight (profiles/r03_concurrent_ppo_hang.log); the library reads the variable once")
        self.is_discrete_actions = is_discrete_actions(self.venv)
        self.is_discrete_adversary_env_actions = is_discrete_actions(self.venv, adversary=True)
        self.agents = {"agent": agent, "adversary_agent": adversary_agent, "adversary_env": adversary_env}
        self.agent_rollout_steps = args.num_steps
        self.adversary_env_rollout_steps = int(self.venv.adversary_observation_space["time_step"].high[0])
        self.is_dr = args.ued_algo == "domain_randomization"
        self.is_training_env = args.ued_algo in ["paired", "flexible_paired", "minimax"]
        self.is_paired = args.ued_algo in ["paired", "flexible_paired"]
        self.requires_batched_vloss = args.use_editor and args.base_levels == "easy"
        if args.ued_algo == "alp_gmm":
            raise NotImplementedError("ALP-GMM only drives BipedalWalker in the reference (SURVEY §2 #21)")
        if args.adv_normalize_returns:
            self.env_return_rms = RunningMeanStd(shape=())
        self.device = torch.device(device)
        self.train() if train else self.eval()
        self.reset()

        self.level_store = None
        self.level_samplers = {}
        self.current_level_seeds = None
        self.weighted_num_edits = 0
        self.latest_env_stats = defaultdict(float)
        if plr_args:
            if self.distributed:
                from level_replay.sharded import ShardedLevelSampler, ShardedLevelStore
                make_sampler = lambda: ShardedLevelSampler(plr_args, group=process_group)
                make_store = lambda **kw: ShardedLevelStore(group=process_group, **kw)
            else:
                make_sampler = lambda: LevelSampler(**plr_args)
                make_store = LevelStore
            if self.is_paired:
                if not args.protagonist_plr and not args.antagonist_plr:
                    self.level_samplers.update({"agent": make_sampler(), "adversary_agent": make_sampler()})
                elif args.protagonist_plr:
                    self.level_samplers["agent"] = make_sampler()
                elif args.antagonist_plr:
                    self.level_samplers["adversary_agent"] = make_sampler()
            else:
                self.level_samplers["agent"] = make_sampler()
            if self.use_byte_encoding:
                example = self.ued_venv.get_encodings()[0]
                self.level_store = make_store(data_info={"numpy": True, "dtype": example.dtype, "shape": example.shape})
            else:
                self.level_store = make_store()
            self.current_level_seeds = [-1 for _ in range(args.num_processes)]
            self._default_level_sampler = self.all_level_samplers[0]
            self.use_editor = args.use_editor
            self.edit_prob = args.level_editor_prob
            self.base_levels = args.base_levels
        else:
            self.use_editor = False
            self.edit_prob = 0
            self.base_levels = None
        B = args.num_processes
        self._done_buf = torch.zeros(B, dtype=torch.uint8, device=self.device)
        self._seeds_dev = torch.full((B, 1), -1, dtype=torch.int32, device=self.device)
        # HIP graphs.  DCD_HIP_GRAPHS=1 (default): student rollouts whose policy step runs on the fused HIP path
        # (dcd_hip.policy: 4 kernel nodes + 1 GEMM node per step) are captured into one graph per rollout and
        # replayed: 17.9 -> 15.5 ms per 256-step rollout at B = 4096.  =2 also captures rollouts whose policy runs
        # as PyTorch modules (teacher, unsupported nets: ~45 nodes per step).
        # =0: always eager.  Replay rollouts that re-sample levels on the host per step always stay eager.
        mode = os.environ.get("DCD_HIP_GRAPHS", "1")
        # DCD_PAIR_ROLLOUT=1 (default): in a PAIRED update without level replay the protagonist's and the antagonist's
        # rollouts advance in lock-step, ONE env-step launch for both (dcd_mg_step_pair) on two copies of the levels.
        self.pair_rollouts = os.environ.get("DCD_PAIR_ROLLOUT", "1") != "0" and self.device.type == "cuda"
        self.pair_policy_launches = self.pair_rollouts and os.environ.get("DCD_PAIR_POLICY", "1") != "0"
        self.fuse_teacher_step = os.environ.get("DCD_FUSED_TEACHER", "1") != "0"
        # DCD_U8_OBS=1 (default): the students' rollout buffers hold the RAW observation bytes (76 B per env-step instead of
        # 304 B of preprocessed fp32); the fused policy step reads them, fp32 views are derived for the PPO batch
        if os.environ.get("DCD_U8_OBS", "1") != "0" and self.device.type == "cuda":
            for name in ("agent", "adversary_agent", "adversary_env"):
                if self.agents[name] is not None:
                    self.agents[name].storage.enable_u8_obs()        # (the teacher's full-grid image: 675 instead of 2 700 B)
        self.use_hip_graphs = mode in ("1", "2") and self.device.type == "cuda"
        self.graph_pytorch_nets = mode == "2"       # also capture rollouts whose policy runs as PyTorch modules
        self._graphs = {}

    # ---- bookkeeping ---------------------------------------------------------------------------
    @property
    def use_byte_encoding(self):
        env_name = self.args.env_name
        return bool(self.args.use_editor or env_name.startswith("BipedalWalker")
                    or (env_name.startswith("MultiGrid") and self.args.use_reset_random_dr))

    def reset(self):
        self.num_updates = 0
        self.total_num_edits = 0
        self.total_episodes_collected = 0
        self.total_seeds_collected = 0
        self.student_grad_updates = 0
        self.sampled_level_info = None
        self.agent_returns = deque(maxlen=10)
        self.adversary_agent_returns = deque(maxlen=10)

    def train(self):
        self.is_training = True
        for a in self.agents.values():
            if a:
                a.train()

    def eval(self):
        self.is_training = False
        for a in self.agents.values():
            if a:
                a.eval()

    def state_dict(self):
        agent_sd, optim_sd = {}, {}
        for k, a in self.agents.items():
            if a:
                agent_sd[k] = a.algo.actor_critic.state_dict()
                optim_sd[k] = a.algo.optimizer.state_dict()
        return {"agent_state_dict": agent_sd, "optimizer_state_dict": optim_sd, "agent_returns": self.agent_returns,
                "adversary_agent_returns": self.adversary_agent_returns, "num_updates": self.num_updates,
                "total_episodes_collected": self.total_episodes_collected,
                "total_seeds_collected": self.total_seeds_collected, "total_num_edits": self.total_num_edits,
                "student_grad_updates": self.student_grad_updates, "latest_env_stats": self.latest_env_stats,
                "level_store": self.level_store, "level_samplers": self.level_samplers}

    def load_state_dict(self, sd):
        for k, state in sd.get("agent_state_dict").items():
            self.agents[k].algo.actor_critic.load_state_dict(state)
        for k, state in sd.get("optimizer_state_dict").items():
            self.agents[k].algo.optimizer.load_state_dict(state)
        for key in ("agent_returns", "adversary_agent_returns", "num_updates", "total_episodes_collected",
                    "total_seeds_collected", "total_num_edits", "student_grad_updates", "latest_env_stats",
                    "level_store", "level_samplers"):
            setattr(self, key, sd.get(key))
        self._table = None                                            # the device mirror of the (replaced) level store refills lazily
        if self.args.use_plr:
            if self.distributed:                                      # checkpoint written by rank 0: re-attach the group
                import torch.distributed as dist
                self.level_store.group = self.process_group
                for s in self.all_level_samplers:
                    s.group = self.process_group
                    s.rank, s.world = dist.get_rank(self.process_group), self.world_size
                    if s.rank != 0:
                        s.sampler = None
            self._default_level_sampler = self.all_level_samplers[0]
            if self.use_editor:
                self.weighted_num_edits = self._get_weighted_num_edits()

    # ---- stats ---------------------------------------------------------------------------------
    def _rollout_return_stats(self, storage, device_stats=None):
        """Per-env mean / max of the episode returns of this rollout (:257-271), on device; plus the
        last <=10 returns in the reference's append order for the running mean (:845-853)."""
        mean, mx, n = device_stats if device_stats is not None else scans.rollout_return_stats(storage.ep_ret, storage.ep_info)
        total = int(n.sum().item())
        tail = []
        # reference order: for b in envs: for r in reversed(returns[b]); the running mean keeps the last 10 appended — they
        # belong to the LAST envs, so only a suffix of the env axis comes to the host (the whole [T, B] pair was 8 MB and
        # a 1M-element nonzero: 5 ms per student and update)
        B, width = storage.ep_info.shape[1], 64
        while total:
            lo = max(0, B - width)
            info = storage.ep_info[:, lo:].t().cpu().numpy()             # [envs, T]
            b_idx, t_idx = np.nonzero(info & dcd_hip.INFO_EPISODE)       # env-major, time ascending
            if len(b_idx) >= 10 or lo == 0:
                rets = storage.ep_ret[:, lo:].t().cpu().numpy()
                order = np.lexsort((-t_idx, b_idx))
                tail = rets[b_idx[order], t_idx[order]][-10:].tolist()
                break
            width *= 8
        return {"mean_return": mean, "max_return": mx, "num_episodes": n, "returns_tail": tail, "total_episodes": total}

    def _get_env_stats_multigrid(self, agent_info, adversary_agent_info):
        num_blocks = np.mean(agent_info.get("num_blocks", self.venv.get_num_blocks()))
        passable_ratio = np.mean(agent_info.get("passable_ratio", self.venv.get_passable()))
        spl = agent_info.get("shortest_path_lengths", self.venv.get_shortest_path_length())
        solved_idx = agent_info.get("solved_idx", None)
        if solved_idx is None:
            solved_idx = (agent_info["max_return"] > 0).cpu().numpy().reshape(-1)
        solved = np.array(spl)[solved_idx]
        return {"num_blocks": num_blocks, "passable_ratio": passable_ratio, "shortest_path_length": np.mean(spl),
                "solved_path_length": np.mean(solved) if len(solved) > 0 else 0}

    def _get_plr_buffer_stats(self):
        stats = {}
        for k, s in self.level_samplers.items():
            stats[k + "_plr_passable_mass"] = self._logged(lambda: s.solvable_mass)
            stats[k + "_plr_max_score"] = s.max_score
            stats[k + "_plr_weighted_num_edits"] = self.weighted_num_edits
        return stats

    @staticmethod
    def _logged(fn):
        """A logging-only read of the replay distribution.  While every buffer slot is still unseen the reference's
        `weights /= np.sum(weights)` (level_sampler.py:736) is 0/0: it warns and logs NaN.  The sampler's arithmetic is kept;
        the LOGGED value of an empty buffer is 0.0 and no warning is raised."""
        with np.errstate(invalid="ignore", divide="ignore"):
            v = fn()
        return 0.0 if isinstance(v, float) and v != v else (0.0 if isinstance(v, np.floating) and np.isnan(v) else v)

    def _get_env_stats(self, agent_info, adversary_agent_info, log_replay_complexity=False):
        if not self.args.env_name.startswith("MultiGrid"):
            raise ValueError(f"Unsupported environment, {self.args.env_name}")
        stats = self._get_env_stats_multigrid(agent_info, adversary_agent_info)
        out = {}
        for k, v in stats.items():
            out["plr_" + k] = v if log_replay_complexity else None
            out[k] = v if not log_replay_complexity else None
        return out

    # ---- PLR plumbing -----------------------------------------------------------------------------
    def _get_active_levels(self):
        assert self.args.use_plr, "Only call _get_active_levels when using PLR."
        if self.use_byte_encoding:
            return [x.tobytes() for x in self.ued_venv.get_encodings()]
        return self.agents["adversary_env"].storage.get_action_traj(as_string=True)

    def _get_level_sampler(self, name):
        other = "agent" if name == "adversary_agent" else "adversary_agent"
        sampler = self.level_samplers.get(name) or self.level_samplers.get(other)
        return sampler, name in self.level_samplers

    @property
    def all_level_samplers(self):
        return [v for v in self.level_samplers.values() if v is not None]

    def _from_rank0(self, value):
        """Host-side random decisions must agree on every rank: rank 0's draw wins (one small tensor broadcast)."""
        if not self.distributed:
            return value
        from util import distributed as D
        return bool(D.broadcast_ints([int(bool(value))], 1, group=self.process_group, device=self.device)[0])

    def _should_edit_level(self):
        return bool(self._from_rank0(bool(self.use_editor and np.random.rand() < self.edit_prob)))

    def _update_plr_with_current_unseen_levels(self, parent_seeds=None):
        levels = self._get_active_levels()
        self.current_level_seeds = self.level_store.insert(levels, parent_seeds=parent_seeds)
        if self.device.type == "cuda":
            if self.use_byte_encoding:
                self._mirror_new_levels(self.current_level_seeds, self.ued_venv.get_encodings_device())
            else:                                            # the teacher's action sequences are still in its storage: [T,B,1] -> [B,T]
                acts = self.agents["adversary_env"].storage.actions[:, :, 0].t().contiguous()
                self._mirror_new_levels(self.current_level_seeds, acts,
                                        torch.full((acts.shape[0],), acts.shape[1], dtype=torch.int32, device=self.device))
        passable = self.venv.get_passable() if (self.args.log_plr_buffer_stats or self.args.reject_unsolvable_seeds) else None
        seeds, solvable = self.current_level_seeds, passable
        if self.args.reject_unsolvable_seeds:
            solvable = np.array(solvable, dtype=bool)
            seeds = np.array(seeds, dtype=int)[solvable]
            solvable = solvable[solvable]
        for sampler in self.all_level_samplers:
            sampler.observe_external_unseen_sample(seeds, solvable)

    def _reconcile_level_store_and_samplers(self):
        all_seeds = set()
        for sampler in self.all_level_samplers:
            all_seeds.update([x for x in sampler.seeds if x >= 0])
        self.level_store.reconcile_seeds(all_seeds)
        self._prune_level_rows()

    def _get_weighted_num_edits(self):
        sampler = self.all_level_samplers[0]
        n = np.zeros(sampler.seed_buffer_size)
        for idx, value in enumerate(self.level_store.seed2parent.values()):
            n[idx] = len(value)
        return self._logged(lambda: np.dot(sampler.sample_weights(), n))

    def _sample_replay_decision(self):
        return self._default_level_sampler.sample_replay_decision()

    # ---- device mirror of the level store ------------------------------------------------------------------------
    # A replay rollout resets envs to stored levels at every step (each finished episode, :551-558) and 4096 at its start;
    # taking the levels from the host store — 675-byte encodings stacked and uploaded, or teacher action strings split,
    # converted and uploaded — was 27 ms of a replay update's 140 ms of per-step host work.  The levels of every seed in use
    # stay in ONE device tensor (uint8 [rows,S,S,3] encodings, or int64 [rows,L] teacher actions + int32 lengths); a reset
    # sends row indices.
    def _level_rows(self, seeds):
        """Rows of `seeds` in the device level table (uploading the ones it does not hold yet, in one batch)."""
        if getattr(self, "_table", None) is None:
            if self.use_byte_encoding:
                S = self.venv.proto.size
                self._table = torch.zeros((0, S, S, 3), dtype=torch.uint8, device=self.device)
            else:
                self._table = torch.zeros((0, max(1, int(self.adversary_env_rollout_steps))), dtype=torch.int64, device=self.device)
                self._table_len = torch.zeros((0,), dtype=torch.int32, device=self.device)
            self._row_of, self._free_rows = {}, []
        row_of = self._row_of
        missing = [s for s in dict.fromkeys(seeds) if s not in row_of]
        if missing:
            kind, a, b = self.venv._levels_to_device([self.level_store.get_level(s) for s in missing])
            if (kind == "enc") != bool(self.use_byte_encoding):
                raise RuntimeError(f"level store holds {kind!r} levels, runner expects use_byte_encoding={self.use_byte_encoding}")
            self._store_rows(missing, a, b)
        return torch.tensor([row_of[s] for s in seeds], dtype=torch.int64, device=self.device)

    def _store_rows(self, seeds, lev_dev, lens_dev=None):
        """lev_dev (device): the levels of `seeds` (n distinct, not yet in the table) — [n,S,S,3] encodings, or [n,L] teacher
        actions with lens_dev [n]."""
        n = len(seeds)
        if lens_dev is not None and lev_dev.shape[1] != self._table.shape[1]:   # action tables: one width (zero padded)
            W = max(lev_dev.shape[1], self._table.shape[1])
            pad = lambda t: torch.cat([t, t.new_zeros((t.shape[0], W - t.shape[1]))], dim=1) if t.shape[1] < W else t
            self._table, lev_dev = pad(self._table), pad(lev_dev)
        while len(self._free_rows) < n:                                   # grow geometrically
            old = self._table.shape[0]
            grow = max(1024, old, n)
            self._table = torch.cat([self._table, self._table.new_zeros((grow,) + tuple(self._table.shape[1:]))])
            if lens_dev is not None:
                self._table_len = torch.cat([self._table_len, self._table_len.new_zeros((grow,))])
            self._free_rows.extend(range(old + grow - 1, old - 1, -1))
        rows = [self._free_rows.pop() for _ in range(n)]
        rows_dev = torch.tensor(rows, dtype=torch.int64, device=self.device)
        self._table[rows_dev] = lev_dev
        if lens_dev is not None:
            self._table_len[rows_dev] = lens_dev.to(torch.int32)
        self._row_of.update(zip(seeds, rows))

    def _mirror_new_levels(self, seeds, lev_dev, lens_dev=None):
        """After level_store.insert: the levels just built are already on the device — keep them (first row per new seed)."""
        if getattr(self, "_table", None) is None:
            self._level_rows([])
        first = {}
        for i, s in enumerate(seeds):
            if s not in self._row_of and s not in first:
                first[s] = i
        if first:
            idx = torch.tensor(list(first.values()), dtype=torch.int64, device=self.device)
            self._store_rows(list(first.keys()), lev_dev.index_select(0, idx), None if lens_dev is None else lens_dev.index_select(0, idx))

    def _prune_level_rows(self):
        if getattr(self, "_table", None) is not None:
            for s in [s for s in self._row_of if s not in self.level_store.seed2level]:
                self._free_rows.append(self._row_of.pop(s))

    def _reset_all_to_stored_levels(self, seeds):
        """venv.reset_to_level_batch([level_store.get_level(s) for s in seeds]) (:457-465)."""
        if self.device.type == "cuda":
            self._reset_to_stored_levels(list(seeds), None, self.venv.engine.student_buffers(u8=False))
            self.ued_venv.raise_errors()
        else:
            self.ued_venv.reset_to_level_batch([self.level_store.get_level(seed) for seed in seeds])

    def _reset_to_stored_levels(self, seeds, idx_dev=None, out=None):
        """reset_to_level for stored seeds (all envs, or the envs `idx_dev`) straight from the device table."""
        if not len(seeds):
            return self.venv.engine.reset_to_encoding(None, idx_dev, out) if self.use_byte_encoding else None
        rows = self._level_rows(seeds)
        if self.use_byte_encoding:
            return self.venv.engine.reset_to_encoding(self._table.index_select(0, rows), idx_dev, out)
        return self.venv.engine.reset_to_actions(self._table.index_select(0, rows), self._table_len.index_select(0, rows), idx_dev, out)

    # ---- rollouts --------------------------------------------------------------------------------
    # Rollout steps per HIP graph.  A whole rollout is ONE graph: a fused 256-step student rollout is ~1.5k kernel nodes, a
    # rollout of PyTorch modules (DCD_HIP_GRAPHS=2) ~45 nodes per step.  The largest capture on record that works on ROCm 7.2
    # is 48 000 kernel nodes; at 80 000 and 120 000 hipStreamEndCapture crashes on the host
    # (profiles/r02_graph_abort_root_cause.log §4, tools/graph_assert_probe.py), so the cap keeps the worst case — a module
    # rollout — below 40k nodes per graph: 800 steps x ~45 nodes = 36k.  (Round 1 chunked to 32 steps after a GPU abort "with
    # 10k-node graphs"; the abort had nothing to do with graph size — see models/distributions.py::FixedCategorical.sample
    # and DESIGN.md §8.)
    GRAPH_CHUNK = 800
    # Captures are thread-local: with a process group in the process, ProcessGroupNCCL's watchdog THREAD polls the events of
    # finished collectives (hipEventQuery) at its own pace; under the default "global" capture mode such a call from any thread
    # while this thread captures aborts the process ("operation not permitted when stream is capturing": the 1-rank RCCL test
    # hit it once the updates between two captures got short enough, round 4).  Nothing another thread of this process does
    # touches the capturing stream (the sampler's prefetch thread makes no HIP call).
    GRAPH_CAPTURE_MODE = "thread_local"

    def _graphed(self, key, num_steps, fn):
        """fn(t0, t1) issues rollout steps [t0, t1).  First call: eager (warm-up); second: captured (one HIP graph per
        GRAPH_CHUNK steps, i.e. one graph) and replayed; afterwards: replay only."""
        if not self.use_hip_graphs:
            return fn(0, num_steps)
        state = self._graphs.get(key)
        if state is None:
            fn(0, num_steps)
            self._graphs[key] = "warm"
            return
        if state == "warm":
            torch.cuda.synchronize()
            state = []
            for t0 in range(0, num_steps, self.GRAPH_CHUNK):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=self.GRAPH_CAPTURE_MODE):
                    fn(t0, min(t0 + self.GRAPH_CHUNK, num_steps))
                state.append(g)
            self._graphs[key] = state
        for g in state:
            g.replay()

    def _teacher_rollout(self, agent, num_steps):
        body = lambda t0, t1: self._teacher_rollout_body(agent, t0, t1)
        ga = self._grid_actor(agent) if self.is_discrete_adversary_env_actions else None
        st = agent.storage
        # the carried projection on raw u8 observations reads nothing back: the whole constructive episode is one graph
        device_resident = (ga is not None and self.use_hip_graphs and st.obs_u8 is not None
                           and ga.device_carry_supported(st.obs_u8["image"][0]))
        if ga is not None:
            ga.begin_rollout(num_steps)                                  # optimiser steps / Philox counters: OUTSIDE the capture
        if self.graph_pytorch_nets or device_resident:
            self._graphed(("teacher", id(agent), num_steps, bool(agent.actor_critic.random)), num_steps, body)
        else:
            body(0, num_steps)
        agent.storage.step = 0
        agent.storage.mark_obs_stale()
        self.ued_venv.raise_errors()
        if ga is not None:
            ga.check_carry_errors()

    def _grid_actor(self, agent):
        """dcd_hip.policy.GridActor for a recurrent teacher (features kernel + one stacked GEMM), or None."""
        ga = getattr(agent, "_grid_actor", False)
        if ga is False:
            from dcd_hip.policy import GridActor
            ga = GridActor.try_create(agent.actor_critic, self.args.num_processes) if self.device.type == "cuda" else None
            if ga is not None:
                ga.seed = (self.args.seed * 2654435761 + 977) & 0xFFFFFFFF
                ga.env_index_base = getattr(self.venv, "env_index_base", 0)
            agent._grid_actor = ga
        if ga is not None and agent.actor_critic.random:
            return None
        return ga

    def _teacher_rollout_body(self, agent, t0, t1):
        """Constructive teacher episode, steps [t0, t1) (:481, :497-592 with is_env=True)."""
        st, venv, B = agent.storage, self.ued_venv, self.args.num_processes
        if t0 == 0:
            venv.reset_device(st.adversary_obs_out(0))
        st.step = t0
        grid_actor = self._grid_actor(agent) if self.is_discrete_adversary_env_actions else None
        if grid_actor is not None:
            grid_actor.refresh()
            if t0 == 0:
                grid_actor.begin_episode()
        st.rewards[t0:t1].zero_()                            # a constructive step pays nothing; no time limits (:497-592 with is_env)
        st.bad_masks[t0 + 1:t1 + 1].fill_(1.0)
        st.cliffhanger_masks[t0 + 1:t1 + 1].fill_(1.0)
        fused_step = (grid_actor is not None and grid_actor.fused_heads and st.obs_u8 is not None and self.fuse_teacher_step
                      and grid_actor._incremental and grid_actor.device_carry_supported(st.obs_u8["image"][0]))
        for step in range(t0, t1):
            if fused_step:                                   # carried projection + rest operand + GEMM + cell / heads / draw -> storage
                grid_actor.act_storage(st, step)
                action = st.actions[step]
            with torch.no_grad():
                if fused_step:
                    pass
                elif grid_actor is not None:
                    img = st.obs_u8["image"] if st.obs_u8 is not None else st._obs["image"]
                    value, action, log_dist, hxs = grid_actor.act(
                        st.rollout_obs(step), st.recurrent_hidden_states[step], st.masks[step], first=step == 0,
                        prev_image=img[step - 1] if step else None)
                else:
                    value, action, log_dist, hxs = agent.act(st.get_obs(step), st.get_recurrent_hidden_state(step),
                                                             st.masks[step])
                if not fused_step:
                    log_prob = log_dist.gather(-1, action) if self.is_discrete_adversary_env_actions else log_dist
            if not fused_step:
                st.insert_policy(hxs, action, log_prob, log_dist, value)
            out = st.adversary_obs_out(step + 1)
            out["done"] = self._done_buf
            venv.step_adversary_device(action.view(-1), out)
            st.masks[step + 1].copy_(self._done_buf.view(B, 1) == 0)      # 1 - done (:520-524); the constant rows: once, above
            st.advance()

    def _resample_finished_levels(self, st, step, level_sampler, rollout_info):
        """PLR replay rollouts: every env that finished an episode gets a new replay level
        (:551-558).  Needs the done mask on the host: one synchronisation per step."""
        finished = (st.ep_info[step] & dcd_hip.INFO_EPISODE).nonzero().flatten()
        idx = finished.cpu().tolist()
        seeds = level_sampler.sample_replay_levels(len(idx))            # a collective when sharded: never skipped
        if not idx:
            return
        idx_dev = finished.to(torch.int32)
        u8 = st.obs_u8 is not None
        out = self._resample_buffers(len(idx), u8)
        self._reset_to_stored_levels(seeds, idx_dev, out)
        if u8:
            st.obs_u8["image"][step + 1][finished] = out["image_u8"]
            st.obs_u8["direction"][step + 1][finished] = out["dir_u8"]
            st._stale[step + 1] = True
        else:
            st._obs["image"][step + 1][finished] = out["image_f32"]
            st._obs["direction"][step + 1][finished] = out["dir_f32"]
        cur = self.current_level_seeds                                   # this rollout's own copy (_student_rollout)
        for i, s in zip(idx, seeds):
            cur[i] = s
        rollout_info["solved_idx"][idx] = True
        self._seeds_dev[finished] = torch.tensor(seeds, dtype=torch.int32).to(self.device, non_blocking=True).view(-1, 1)

    def _resample_buffers(self, n, u8):
        """Observation outputs for n re-sampled envs: slices of buffers allocated once (a replay rollout asks at every step)."""
        key = "u8" if u8 else "f32"
        bufs = getattr(self, "_resample_out", None)
        if bufs is None:
            bufs = self._resample_out = {}
        if key not in bufs:
            bufs[key] = self.venv.engine.student_buffers(self.args.num_processes, u8=u8, f32=not u8)
        return {k: v[:n] for k, v in bufs[key].items()}

    def _student_rollout(self, agent, num_steps, level_replay, level_sampler, rollout_info):
        track_seeds = level_sampler is not None
        if track_seeds:
            self._seeds_dev.copy_(torch.tensor(self.current_level_seeds, dtype=torch.int32).view(-1, 1))
        host_resampling = level_sampler is not None and level_replay
        if host_resampling:                     # per-step re-sampling edits the seed list in place; the list sampled at the
            self.current_level_seeds = list(self.current_level_seeds)   # start of the rollout stays with its other holders
        fused = self._fused_actor(agent) if self.is_discrete_actions else None
        if fused is not None:
            fused.begin_rollout(num_steps)
        body = lambda t0, t1: self._student_rollout_body(agent, num_steps, t0, t1, level_replay, level_sampler, rollout_info)
        can_graph = self.use_hip_graphs and (fused is not None or self.graph_pytorch_nets)
        if (host_resampling and can_graph and fused is not None and self.replay_queue
                and self._queued_replay_rollout(agent, num_steps, level_sampler, rollout_info)):
            pass                                                        # the whole rollout ran as one graph on queued draws
        elif host_resampling and can_graph:
            # Replay rollouts re-sample a level for every finished episode ON THE HOST between two steps (:551-558): the
            # device part of a step (policy, env step, bookkeeping) is one small HIP graph per step index, replayed, then
            # the host looks at the finished mask.  (Eager launches made this rollout 120 ms against 16 ms of GPU work.)
            key = ("student-steps", id(agent), num_steps, bool(agent.actor_critic.random))
            graphs = self._graphs.get(key)
            dev_step = lambda t: self._student_rollout_body(agent, num_steps, t, t + 1, False, level_sampler, rollout_info)
            if graphs is None:                                          # first replay rollout: eager, warms everything up
                self._graphs[key] = "warm"
                body(0, num_steps)
            else:
                if graphs == "warm":
                    torch.cuda.synchronize()
                    graphs = []
                    for t in range(num_steps):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, capture_error_mode=self.GRAPH_CAPTURE_MODE):
                            dev_step(t)
                        graphs.append(g)
                    self._graphs[key] = graphs
                st = agent.storage
                for t in range(num_steps):
                    graphs[t].replay()
                    self._resample_finished_levels(st, t, level_sampler, rollout_info)
        elif host_resampling or not can_graph:
            body(0, num_steps)
        else:
            self._graphed(("student", id(agent), num_steps, track_seeds, bool(agent.actor_critic.random)), num_steps, body)
        agent.storage.step = 0
        agent.storage.mark_obs_stale()
        self.venv.raise_errors()

    # ---- replay rollouts on a device-resident queue of draws ---------------------------------------------------------------
    # adversarial_runner.py:551-558 draws a replay level for every finished env, in env order, between two steps.  Those draws
    # are one sequential stream of the sampler (level_sampler.py:664-680) that the host has already speculated ahead while the
    # GPU ran the previous PPO update (LevelSampler.prefetch_next_update).  Its next entries — seed and level-table row — are
    # uploaded once; at every step `dcd_replay_pop` hands them to the finished envs in env order and a masked level reset
    # (dcd_mg_reset_to_*_rows) loads them, observations landing in the rollout buffers in place: the rollout is ONE graph
    # with one synchronisation at its end.  The host then pops exactly the consumed prefix from the sampler
    # (sample_replay_levels(consumed): same seeds, same np.random consumption, same staleness as 256 per-step requests — the
    # stream does not depend on how requests cut it).  If the queue runs dry (more finished episodes than speculated draws) the
    # kernel raises a flag: the env batch is restored from the snapshot taken before the rollout (the twin replica) and the
    # rollout is redone on the per-step host path — same result, deterministically.
    def _queued_replay_rollout(self, agent, num_steps, level_sampler, rollout_info):
        import ctypes as C
        B, st, venv, dev = self.args.num_processes, agent.storage, self.venv, self.device
        if not hasattr(level_sampler, "peek_stream") or venv.engine.cfg.replicas < 2:
            return False
        kmax = self.prefetch_stream_factor * B
        queue = level_sampler.peek_stream(min(B, kmax), kmax)
        if not queue:
            return False
        q = getattr(self, "_replay_q", None)
        if q is None or q["rows"].numel() != kmax:
            i32 = dict(dtype=torch.int32, device=dev)
            q = self._replay_q = dict(rows=torch.full((kmax,), -1, **i32), seeds=torch.zeros((kmax,), **i32), state=torch.zeros((2,), **i32),
                                      level_idx=torch.full((B,), -1, **i32), solved=torch.zeros((B,), dtype=torch.uint8, device=dev))
        rows = self._level_rows(queue)                                   # uploads what the table lacks (may reallocate it)
        q["rows"].fill_(-1)
        q["rows"][:len(queue)].copy_(rows.to(torch.int32))
        q["seeds"][:len(queue)].copy_(torch.tensor(queue, dtype=torch.int32), non_blocking=True)
        q["state"].zero_()                                               # [cursor, overflow]
        q["solved"].zero_()
        venv.clone_to_twin()                                             # snapshot: records, Philox keys / counters, metrics
        seeds_before = self._seeds_dev.clone()
        fused = self._fused_actor(agent)
        L = dcd_hip.lib()
        table, lens = self._table, getattr(self, "_table_len", None)

        def body(t0, t1):
            for t in range(t0, t1):
                self._student_rollout_body(agent, num_steps, t, t + 1, False, level_sampler, rollout_info)
                dcd_hip.check(L.dcd_replay_pop(dcd_hip.ptr(st.ep_info[t], torch.uint8, B, "ep_info"), C.c_int32(B), dcd_hip.ptr(q["rows"]),
                                               dcd_hip.ptr(q["seeds"]), C.c_int32(kmax), C.c_void_p(q["state"].data_ptr()),
                                               C.c_void_p(q["state"].data_ptr() + 4), dcd_hip.ptr(q["level_idx"]),
                                               dcd_hip.ptr(self._seeds_dev, torch.int32, B, "seeds"), dcd_hip.ptr(q["solved"]), dcd_hip.stream_ptr()))
                venv.engine.reset_to_table_rows(table, lens, q["level_idx"], st.student_obs_out(t + 1))
        # the graph holds the level table's address: a reallocated table (grown by _store_rows) gets a new graph
        key = ("student-queue", id(agent), num_steps, bool(agent.actor_critic.random), table.data_ptr(), 0 if lens is None else lens.data_ptr())
        for k in [k for k in self._graphs if k[0] == "student-queue" and k[1] == id(agent) and k != key]:
            del self._graphs[k]
        self._graphed(key, num_steps, body)
        consumed, overflow = q["state"].cpu().tolist()                   # the rollout's one synchronisation
        if overflow:
            venv.engine.clone_levels(venv.twin)                          # roll the env batch back; the host path redoes the rollout
            self._seeds_dev.copy_(seeds_before)
            self.replay_queue_overflows = getattr(self, "replay_queue_overflows", 0) + 1
            return False
        seeds = level_sampler.sample_replay_levels(consumed)
        if list(seeds) != queue[:consumed]:
            raise RuntimeError("replay queue: the sampler handed out other draws than the ones that were queued")
        self.current_level_seeds[:] = self._seeds_dev.view(-1).cpu().tolist()
        solved = q["solved"].cpu().numpy().astype(bool)
        rollout_info["solved_idx"][solved] = True
        self.replay_queue_rollouts = getattr(self, "replay_queue_rollouts", 0) + 1
        return True

    def _fused_actor(self, agent):
        """dcd_hip.policy.StudentActor for this agent's network (act + insert_policy as two HIP kernels and one
        GEMM), or None when the network is not student-shaped / DCD_FUSED_ACT=0: then the PyTorch modules run."""
        fa = getattr(agent, "_fused_actor", False)
        if fa is False:
            from dcd_hip.policy import PairedStudentActors, StudentActor, enabled
            live = [a for a in self.agents.values() if a is not None]
            students = [self.agents.get("agent"), self.agents.get("adversary_agent")]
            group, buf = [agent], None
            # PAIRED: both students' actors are created together on the halves of shared [2, ...] operand / weight tensors, so
            # that a lock-step rollout runs ONE features launch, ONE batched GEMM and ONE cell + heads launch per step for
            # both (the buffers are fixed before any graph is captured)
            if (self.device.type == "cuda" and enabled() and self.pair_policy_launches and all(a is not None for a in students)
                    and any(a is agent for a in students)
                    and all(StudentActor.supported(a.actor_critic) and getattr(a, "_fused_actor", False) is False for a in students)):
                buf = PairedStudentActors.buffers([a.actor_critic for a in students], self.args.num_processes)
                if buf is not None:
                    group = students
            made = []
            for j, ag in enumerate(group):
                k = live.index(ag) if ag in live else 0
                f = None
                if self.device.type == "cuda":
                    f = StudentActor.try_create(ag.actor_critic, self.args.num_processes,
                                                seed=(self.args.seed * 2654435761 + 97 * k + 1) & 0xFFFFFFFF,
                                                env_index_base=getattr(self.venv, "env_index_base", 0),
                                                buffers=PairedStudentActors.half(buf, j) if buf is not None and len(group) == 2 else None)
                ag._fused_actor = f
                made.append(f)
            if len(group) == 2 and all(f is not None for f in made):
                self._pair_actors = PairedStudentActors(made, buf)
            fa = agent._fused_actor
        if fa is not None and agent.actor_critic.random:
            return None
        return fa

    def _bootstrap_value(self, agent):
        """The value of the observation after the last step (adversarial_runner.py:586-590), by the fused policy step
        when the rollout used it (3 launches, no fp32 observation materialised), else by the PyTorch modules."""
        st = agent.storage
        fused = self._fused_actor(agent) if self.is_discrete_actions else None
        if fused is not None and not st.use_popart:
            return fused.value_storage(st, st.num_steps)
        with torch.no_grad():
            return agent.get_value(st.get_obs(-1), st.get_recurrent_hidden_state(-1), st.masks[-1]).detach()

    def _student_rollout_body(self, agent, num_steps, t0, t1, level_replay, level_sampler, rollout_info):
        args, st, venv = self.args, agent.storage, self.venv
        reset_random = self.is_dr and not args.use_plr
        if t0 == 0:
            venv.reset_agent_device(st.student_obs_out(0))
        st.step = t0
        track_seeds = level_sampler is not None
        fused = self._fused_actor(agent) if self.is_discrete_actions else None
        for step in range(t0, t1):
            if fused is not None:
                fused.act_storage(st, step)
                if track_seeds:
                    st.level_seeds[step].copy_(self._seeds_dev)
                action = st.actions[step]
            else:
                with torch.no_grad():
                    value, action, log_dist, hxs = agent.act(st.get_obs(step), st.get_recurrent_hidden_state(step),
                                                             st.masks[step])
                    log_prob = log_dist.gather(-1, action) if self.is_discrete_actions else log_dist
                st.insert_policy(hxs, action, log_prob, log_dist, value, self._seeds_dev if track_seeds else None)
            last = step >= num_steps - 1
            venv.step_env_device(action.view(-1), st.step_out(step), reset_random=reset_random,
                                 last_step=last and st.use_proper_time_limits)
            if last and not st.use_proper_time_limits:
                st.masks[step + 1].zero_()                              # done forced on the last step (:530)
            if args.clip_reward:
                st.rewards[step].clamp_(-args.clip_reward, args.clip_reward)
            if level_sampler is not None and level_replay:
                self._resample_finished_levels(st, step, level_sampler, rollout_info)
            st.advance()

    def _paired_student_rollouts(self, agents, num_steps, track_seeds):
        """The protagonist's and the antagonist's rollouts of one update (adversarial_runner.py:709-733: two independent
        `agent_rollout` calls on the teacher's levels, each starting with reset_agent :484) in lock-step: per step both
        policies act, then ONE launch advances both copies of the envs.  Requires levels made of walls and a goal
        (what the teacher builds): then the second rollout sees exactly the level the first one started from, as it does
        in the reference, where reset_agent restores the agent and nothing else can have changed."""
        if track_seeds:
            self._seeds_dev.copy_(torch.tensor(self.current_level_seeds, dtype=torch.int32).view(-1, 1))
        fused = [self._fused_actor(a) if self.is_discrete_actions else None for a in agents]
        for f in fused:
            if f is not None:
                f.begin_rollout(num_steps)
        body = lambda t0, t1: self._paired_rollout_body(agents, num_steps, t0, t1, track_seeds)
        if all(f is not None for f in fused) or self.graph_pytorch_nets:
            key = ("pair", id(agents[0]), id(agents[1]), num_steps, track_seeds) + tuple(bool(a.actor_critic.random) for a in agents)
            self._graphed(key, num_steps, body)
        else:
            body(0, num_steps)
        for a in agents:
            a.storage.step = 0
            a.storage.mark_obs_stale()
        self.venv.raise_errors()
        # both students' return statistics — and, when training, their returns — from one launch each (the *_batch scans)
        B = self.args.num_processes
        outs = [(torch.empty((B, 1), device=self.device), torch.empty((B, 1), device=self.device),
                 torch.empty((B,), dtype=torch.int32, device=self.device)) for _ in agents]
        stat_sets = [(a.storage.ep_ret, a.storage.ep_info) + o for a, o in zip(agents, outs)]
        self._pair_cache = {id(a): {"stats": o} for a, o in zip(agents, outs)}
        if self.is_training:
            nvs = [self._bootstrap_value(a) for a in agents]
            type(agents[0].storage).compute_returns_batch([a.storage for a in agents], nvs, self.args.use_gae, self.args.gamma,
                                                          self.args.gae_lambda, stat_sets=stat_sets)
            for a in agents:
                self._pair_cache[id(a)]["returns"] = True
        else:
            scans.rollout_return_stats_batch(stat_sets)

    def _paired_rollout_body(self, agents, num_steps, t0, t1, track_seeds):
        args, venv = self.args, self.venv
        sts = [a.storage for a in agents]
        if t0 == 0:
            venv.clone_to_twin()
            venv.reset_agent_device(sts[0].student_obs_out(0))
            venv.reset_agent_twin_device(sts[1].student_obs_out(0))
        fused = [self._fused_actor(a) if self.is_discrete_actions else None for a in agents]
        pa = getattr(self, "_pair_actors", None)
        both = (pa is not None and all(f is not None for f in fused) and all(f is g for f, g in zip(fused, pa.actors))
                and all(st.obs_u8 is not None for st in sts))
        for st in sts:
            st.step = t0
        for step in range(t0, t1):
            if both:                                                    # one launch of each policy kernel for both students
                pa.act_storage(sts, step)
                if track_seeds:
                    for st in sts:
                        st.level_seeds[step].copy_(self._seeds_dev)
            for agent, st, f in zip(agents, sts, fused) if not both else ():
                if f is not None:
                    f.act_storage(st, step)
                    if track_seeds:
                        st.level_seeds[step].copy_(self._seeds_dev)
                else:
                    with torch.no_grad():
                        value, action, log_dist, hxs = agent.act(st.get_obs(step), st.get_recurrent_hidden_state(step), st.masks[step])
                        log_prob = log_dist.gather(-1, action) if self.is_discrete_actions else log_dist
                    st.insert_policy(hxs, action, log_prob, log_dist, value, self._seeds_dev if track_seeds else None)
            last = step >= num_steps - 1
            venv.step_env_pair_device(sts[0].actions[step].view(-1), sts[1].actions[step].view(-1), sts[0].step_out(step),
                                      sts[1].step_out(step), last_step=last and sts[0].use_proper_time_limits)
            for st in sts:
                if last and not st.use_proper_time_limits:
                    st.masks[step + 1].zero_()                          # done forced on the last step (:530)
                if args.clip_reward:
                    st.rewards[step].clamp_(-args.clip_reward, args.clip_reward)
                st.advance()

    def agent_rollout(self, agent, num_steps, update=False, is_env=False, level_replay=False, level_sampler=None,
                      update_level_sampler=False, discard_grad=False, edit_level=False, num_edits=0, fixed_seeds=None,
                      rollout_done=False):
        """rollout_done: the storage was already filled by `_paired_student_rollouts`; only the bookkeeping runs."""
        args = self.args
        B = args.num_processes
        if is_env:
            if edit_level:
                self._reset_all_to_stored_levels(fixed_seeds)
                self.ued_venv.mutate_level(num_edits=num_edits)
                self._update_plr_with_current_unseen_levels(parent_seeds=fixed_seeds)
                return
            if level_replay:
                self.current_level_seeds = level_sampler.sample_replay_levels(B)
                self._reset_all_to_stored_levels(self.current_level_seeds)
                return self.current_level_seeds
            if self.is_dr and not args.use_plr:
                self.ued_venv.reset_random_device({})
                self.total_seeds_collected += B
                return
            if self.is_dr and args.use_plr and args.use_reset_random_dr:
                self.ued_venv.reset_random_device({})
                self._update_plr_with_current_unseen_levels(parent_seeds=fixed_seeds)
                self.total_seeds_collected += B
                return
            self.total_seeds_collected += B
            self._teacher_rollout(agent, num_steps)
            if args.use_plr and not level_replay:
                self._update_plr_with_current_unseen_levels()
            return {}

        rollout_info = {}
        if any(a is agent for a, _, _ in self._pending_updates):          # ACCEL: the same student acts again -> new weights first
            self._flush_updates()
        if level_sampler and level_replay:
            rollout_info["solved_idx"] = np.zeros(B, dtype=bool)
        cache = {}
        if not rollout_done:
            self._student_rollout(agent, num_steps, level_replay, level_sampler, rollout_info)
        else:
            cache = getattr(self, "_pair_cache", {}).pop(id(agent), {})
        stats = self._rollout_return_stats(agent.storage, cache.get("stats"))
        rollout_info.update(stats)
        self.total_episodes_collected += stats["total_episodes"]
        if self.is_dr and not args.use_plr:
            self.total_seeds_collected += stats["total_episodes"]
        if update:
            st = agent.storage
            if not cache.get("returns"):
                st.compute_returns(self._bootstrap_value(agent), args.use_gae, args.gamma, args.gae_lambda)
            if self.requires_batched_vloss:
                clipped = not args.adv_use_popart and not args.adv_normalize_returns
                rollout_info["batched_value_loss"] = st.get_batched_value_loss(
                    signed=False, positive_only=False, clipped=clipped, batched=True)
            if level_sampler and update_level_sampler:
                level_sampler.update_with_rollouts(st)
                level_sampler.after_update()                              # sampler bookkeeping: independent of the PPO result
                # the sampler now is what the NEXT update's replay decision and first draws will see: they are computed on a
                # host thread (C loop, outside the GIL) underneath this update's PPO kernels and used if nothing moves in between
                if (self.prefetch_draws and self.is_training and len(self.all_level_samplers) == 1 and not self.use_editor
                        and level_sampler is self._default_level_sampler):
                    # + the stream of per-episode re-sampling draws a replay rollout will ask for step by step (~1.5 per env
                    # at 250-step episodes; what is not speculated is drawn in program order)
                    level_sampler.prefetch_next_update(args.num_processes, stream_extra=self.prefetch_stream_factor * args.num_processes)
            self._submit_update(agent, rollout_info, discard_grad=discard_grad)
            if args.log_action_complexity:
                if self._concurrent_updates:              # host-side LZ76 pass (5 ms per student): runs under the PPO updates
                    self._deferred_host.append((rollout_info, st, st.action_complexity_inputs()))
                else:
                    rollout_info["action_complexity"] = st.get_action_complexity()
        return rollout_info

    # ---- PPO updates of one run(): independent of each other, issued together ---------------------------------
    def _submit_update(self, agent, info, discard_grad=False):
        """`agent.update()` (adversarial_runner.py:600-606,810-812) with its losses written into `info`.  The
        protagonist's, the antagonist's and the teacher's updates of one run() touch disjoint networks and storages,
        and nothing between a student's rollout and the end of run() reads a student's new weights (ACCEL's second
        rollout does: agent_rollout flushes first).  With DCD_CONCURRENT_PPO=1 on one GPU they are therefore queued here and
        issued together by `_flush_updates`, each from its own host thread on its own stream: one update is a serial chain of
        small GEMMs and pointwise kernels (the LSTM recurrence), and three independent chains fill the machine better than one.
        Round 3: OFF by default — with the library's default (StreamK) GEMMs concurrent streams can deadlock on the GPU; with
        data-parallel GEMMs (the only safe setting) the three updates take 0.58 s against 0.597 s in program order
        (profiles/r03_concurrent_ppo_hang.log), a gain that does not pay for a process-wide BLAS override.  With several
        ranks the updates stay in program order: their gradient all-reduces must be issued in the same order on every rank."""
        self._pending_updates.append((agent, info, discard_grad))
        if not self._concurrent_updates:
            self._flush_updates()

    def _run_deferred_host(self):
        todo, self._deferred_host = self._deferred_host, []
        for info, st, inputs in todo:
            info["action_complexity"] = st.get_action_complexity(inputs)

    def _flush_updates(self):
        jobs, self._pending_updates = self._pending_updates, []
        if not jobs:
            self._run_deferred_host()
            return

        def one(agent, info, discard_grad):
            v, a, e, extra = agent.update(discard_grad=discard_grad)
            info.update({"value_loss": v, "action_loss": a, "dist_entropy": e, "update_info": extra})

        if len(jobs) == 1 or self.device.type != "cuda":
            for job in jobs:
                one(*job)
            self._run_deferred_host()
            return
        import threading
        for agent, _, _ in jobs:                  # minibatch permutations in program order, whatever the threads do later
            agent.storage.predraw_permutations(agent.algo.ppo_epoch, agent.is_recurrent)
        main = torch.cuda.current_stream(self.device)
        errors = []

        def work(job, stream):
            try:
                stream.wait_stream(main)                                  # rollouts, returns: produced on the caller's stream
                with torch.cuda.stream(stream):
                    one(*job)
            except BaseException as exc:                                  # re-raised on the caller's thread
                errors.append(exc)

        while len(self._update_streams) < len(jobs):
            self._update_streams.append(torch.cuda.Stream(self.device))
        threads = [threading.Thread(target=work, args=(job, st)) for job, st in zip(jobs, self._update_streams)]
        # Safe only with data-parallel GEMMs (checked at construction): every fp32 GEMM of this stack (ROCm 7.2 hipBLASLt /
        # Tensile on gfx950) is a StreamK kernel — workgroups that share an output tile hand partial sums over through flags and
        # SPIN on them; with GEMMs of several streams in flight each kernel can hold compute units while it waits for workgroups
        # of its own that cannot be scheduled.  Reproduced: PAIRED at 4096 envs x 8 steps hangs in its first concurrent update,
        # every time, in the backward passes of two of the three streams; under TENSILE_STREAMK_DATA_PARALLEL=1 (whole tiles per
        # workgroup, no hand-over; read once per process) 30 such updates run (tools/b4096_bisect.sh).
        for t in threads:
            t.start()
        self._run_deferred_host()                                         # host work of this thread while the GPU trains
        for t in threads:
            t.join()
        for st in self._update_streams[:len(jobs)]:
            main.wait_stream(st)
        if errors:
            raise errors[0]

    def _compute_env_return(self, agent_info, adversary_agent_info):
        """Teacher's final return: the regret (:637-674), on device."""
        args = self.args
        mode = {"paired": 0, "flexible_paired": 1, "minimax": 2}.get(args.ued_algo)
        if mode is None:
            env_return = torch.zeros_like(agent_info["mean_return"])
        else:
            adv = adversary_agent_info if self.is_paired else agent_info
            env_return = scans.env_return(agent_info["mean_return"], agent_info["max_return"], adv["mean_return"],
                                          adv["max_return"], mode=mode)
        if args.adv_normalize_returns:
            sample = env_return.flatten()
            if self.distributed:                    # moments of the whole job's batch, in global env order
                import torch.distributed as dist
                parts = [torch.empty_like(sample) for _ in range(self.world_size)]
                dist.all_gather(parts, sample.contiguous(), group=self.process_group)
                sample = torch.cat(parts)
            self.env_return_rms.update(sample.cpu().numpy())
            env_return = env_return / float(np.sqrt(self.env_return_rms.var + 1e-8))
        if args.adv_clip_reward is not None:
            env_return = env_return.clamp(-args.adv_clip_reward, args.adv_clip_reward)
        return env_return

    # ---- one update ------------------------------------------------------------------------------
    def run(self):
        args = self.args
        adversary_env, agent, adversary_agent = (self.agents[k] for k in ("adversary_env", "agent", "adversary_agent"))
        level_replay = bool(args.use_plr and self.is_training and self._sample_replay_decision())
        student_discard_grad = bool(args.use_plr and (not level_replay)
                                    and vars(args).get("no_exploratory_grad_updates", False))
        if self.is_training and not student_discard_grad:
            self.student_grad_updates += 1

        env_info = self.agent_rollout(agent=adversary_env, num_steps=self.adversary_env_rollout_steps, update=False,
                                      is_env=True, level_replay=level_replay,
                                      level_sampler=self._get_level_sampler("agent")[0], update_level_sampler=False)
        level_sampler, updateable = self._get_level_sampler("agent")
        # plain PAIRED update: both students' rollouts in lock-step on two copies of the levels, then each one's bookkeeping
        paired = bool(self.is_paired and not level_replay and self.pair_rollouts and self.is_discrete_actions
                      and not getattr(self.venv, "normalize_returns", False)     # one running-return state per venv: keep the order
                      and agent.storage.use_proper_time_limits == adversary_agent.storage.use_proper_time_limits)
        if paired:
            if any(a is ag for ag in (agent, adversary_agent) for a, _, _ in self._pending_updates):
                self._flush_updates()
            self._paired_student_rollouts((agent, adversary_agent), self.agent_rollout_steps, track_seeds=bool(self.level_samplers))
        agent_info = self.agent_rollout(agent=agent, num_steps=self.agent_rollout_steps, update=self.is_training,
                                        level_replay=level_replay, level_sampler=level_sampler,
                                        update_level_sampler=updateable, discard_grad=student_discard_grad,
                                        rollout_done=paired)
        if level_replay and self.is_paired and (args.protagonist_plr == args.antagonist_plr):
            self.agent_rollout(agent=adversary_env, num_steps=self.adversary_env_rollout_steps, update=False,
                               is_env=True, level_replay=level_replay,
                               level_sampler=self._get_level_sampler("adversary_agent")[0], update_level_sampler=False)
        adversary_agent_info = defaultdict(float)
        if self.is_paired:
            level_sampler, updateable = self._get_level_sampler("adversary_agent")
            adversary_agent_info = self.agent_rollout(agent=adversary_agent, num_steps=self.agent_rollout_steps,
                                                      update=self.is_training, level_replay=level_replay,
                                                      level_sampler=level_sampler, update_level_sampler=updateable,
                                                      discard_grad=student_discard_grad, rollout_done=paired)

        edit_level = self._should_edit_level() and level_replay
        if level_replay:
            sampled_level_info = {"level_replay": True,
                                  "num_edits": [len(self.level_store.seed2parent[x]) + 1 for x in env_info]}
        else:
            sampled_level_info = {"level_replay": False, "num_edits": [0 for _ in range(args.num_processes)]}

        if level_replay and edit_level:                                   # ---- ACCEL (:758-794)
            if self.base_levels == "batch":
                fixed_seeds = env_info
            elif self.base_levels == "easy":
                ease = (agent_info["mean_return"] - agent_info["batched_value_loss"]).detach().cpu().numpy().flatten()
                if args.num_processes >= 4:
                    easy = list(np.argsort(ease)[:4])
                    fixed_seeds = [env_info[int(x)] for x in easy] * int(args.num_processes / 4)
                else:
                    fixed_seeds = [env_info[int(np.argmax(ease))]] * args.num_processes
            level_sampler, updateable = self._get_level_sampler("agent")
            self.agent_rollout(agent=None, num_steps=None, is_env=True, edit_level=True, num_edits=args.num_edits,
                               fixed_seeds=fixed_seeds)
            self.total_num_edits += 1
            sampled_level_info["num_edits"] = [x + 1 for x in sampled_level_info["num_edits"]]
            self.agent_rollout(agent=agent, num_steps=self.agent_rollout_steps, update=self.is_training,
                               level_replay=False, level_sampler=level_sampler, update_level_sampler=updateable,
                               discard_grad=True)

        if args.use_plr:
            self._reconcile_level_store_and_samplers()
            if self.use_editor:
                self.weighted_num_edits = self._get_weighted_num_edits()

        env_return = self._compute_env_return(agent_info, adversary_agent_info)
        adversary_env_info = defaultdict(float)
        if self.is_training and self.is_training_env:
            st = adversary_env.storage
            with torch.no_grad():
                next_value = adversary_env.get_value(st.get_obs(-1), st.get_recurrent_hidden_state(-1),
                                                     st.masks[-1]).detach()
            st.replace_final_return(env_return)
            st.compute_returns(next_value, args.use_gae, args.gamma, args.gae_lambda)
            self._submit_update(adversary_env, adversary_env_info)
        self._flush_updates()
        if self.is_training:
            self.num_updates += 1

        # ---- logging (:825-896) ----------------------------------------------------------------
        log_replay_complexity = level_replay and args.log_replay_complexity
        if (not level_replay) or log_replay_complexity:
            stats = self._get_env_stats(agent_info, adversary_agent_info, log_replay_complexity=log_replay_complexity)
            stats.update({"mean_env_return": env_return.mean().item(),
                          "adversary_env_pg_loss": adversary_env_info["action_loss"],
                          "adversary_env_value_loss": adversary_env_info["value_loss"],
                          "adversary_env_dist_entropy": adversary_env_info["dist_entropy"]})
            if args.use_plr:
                self.latest_env_stats.update(stats)
        else:
            stats = self.latest_env_stats.copy()
        if args.use_plr and args.log_plr_buffer_stats:
            stats.update(self._get_plr_buffer_stats())
        self.agent_returns.extend(agent_info["returns_tail"])
        mean_agent_return = np.mean(self.agent_returns) if len(self.agent_returns) > 0 else 0
        mean_adversary_agent_return = 0
        if self.is_paired:
            self.adversary_agent_returns.extend(adversary_agent_info["returns_tail"])
            if len(self.adversary_agent_returns) > 0:
                mean_adversary_agent_return = np.mean(self.adversary_agent_returns)
        self.sampled_level_info = sampled_level_info
        stats.update({
            "steps": (self.num_updates + self.total_num_edits) * args.num_processes * args.num_steps,
            "total_episodes": self.total_episodes_collected, "total_seeds": self.total_seeds_collected,
            "total_student_grad_updates": self.student_grad_updates, "mean_agent_return": mean_agent_return,
            "agent_value_loss": agent_info["value_loss"] if "value_loss" in agent_info else 0.0,
            "agent_pg_loss": agent_info.get("action_loss", 0.0), "agent_dist_entropy": agent_info.get("dist_entropy", 0.0),
            "mean_adversary_agent_return": mean_adversary_agent_return,
            "adversary_value_loss": adversary_agent_info["value_loss"],
            "adversary_pg_loss": adversary_agent_info["action_loss"],
            "adversary_dist_entropy": adversary_agent_info["dist_entropy"]})
        if args.log_grad_norm:
            stats.update({
                "agent_grad_norm": np.mean(agent_info["update_info"]["grad_norms"]),
                "adversary_grad_norm": np.mean(adversary_agent_info["update_info"]["grad_norms"]) if self.is_paired else 0,
                "adversary_env_grad_norm": np.mean(adversary_env_info["update_info"]["grad_norms"]) if self.is_training_env else 0})
        if args.log_action_complexity:
            stats.update({"agent_action_complexity": agent_info["action_complexity"],
                          "adversary_action_complexity": adversary_agent_info["action_complexity"]})
        return stats
