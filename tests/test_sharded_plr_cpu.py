"""CPU, world_size 2 and 8 over gloo: level replay with the env batch sharded over ranks.

One LevelSampler on rank 0 fed by all-gathered episode records + a replicated LevelStore must behave
exactly like the reference's single process holding all world x B envs: same replay decisions, same seeds
handed to every env, same sampler / store state after several robust-PLR updates (staging -> working buffer,
eviction, staleness, solvable tracking, per-step re-sampling with ragged per-rank counts).
"""
import os
import pickle
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "dcd-isaac_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

REC = np.dtype([("actor", "<i4"), ("seed", "<i4"), ("start_t", "<i4"), ("num_steps", "<i4"), ("mean_score", "<f4"),
                ("max_score", "<f4"), ("reward_sum", "<f4"), ("flags", "<i4")])
B_LOCAL, UPDATES = 4, 8


class _Space:
    n = 7


def _plr_args(num_actors):
    return dict(seeds=[], obs_space=None, action_space=_Space(), num_actors=num_actors, strategy="positive_value_loss",
                replay_schedule="proportionate", score_transform="rank", temperature=0.3, rho=0.5, replay_prob=0.6,
                alpha=1.0, staleness_coef=0.3, staleness_transform="power", staleness_temperature=1.0,
                sample_full_distribution=True, seed_buffer_size=6, seed_buffer_priority="replay_support", gamma=0.995)


def _records(u, g0, seeds):
    """Synthetic records of envs [g0, g0+len(seeds)) for update u: 1-2 finished episodes + a trailing partial."""
    out = []
    for i, seed in enumerate(seeds):
        g = g0 + i
        rng = np.random.default_rng(1000 * u + g)
        t = 0
        for _ in range(int(rng.integers(1, 3))):
            n = int(rng.integers(2, 6))
            flags = 1 | (2 if rng.random() < 0.15 else 0)
            out.append((i, seed, t, n, rng.random(), rng.random() + 1, rng.random(), flags))
            t += n
        out.append((i, seed, t, int(rng.integers(1, 4)), rng.random(), rng.random() + 1, 0.0, 0))
    return np.array(out, dtype=REC)


def _drive(sampler, store, g0, n_local, update_records, prefetch=False):
    """The runner's PLR call sequence (adversarial_runner.py:676-896) over `n_local` envs starting at global env g0."""
    log = []
    for u in range(UPDATES):
        replay = sampler.sample_replay_decision()
        if replay:
            seeds = sampler.sample_replay_levels(n_local)
        else:
            levels = [f"lvl-{u}-{(g0 + i) % 5}" for i in range(n_local)]
            seeds = store.insert(levels)
            sampler.observe_external_unseen_sample(seeds, [(g0 + i + u) % 3 != 0 for i in range(n_local)])
        if replay:                                                   # per-step re-sampling, ragged counts per rank
            for step in range(3):
                n_fin = sum(1 for i in range(n_local) if (g0 + i + step + u) % 4 == 0)
                new = sampler.sample_replay_levels(n_fin)
                k = 0
                for i in range(n_local):
                    if (g0 + i + step + u) % 4 == 0:
                        seeds[i] = new[k]
                        k += 1
        update_records(_records(u, g0, seeds))
        sampler.after_update()
        if prefetch:                                                 # as the runner: the next update's first draws, speculatively
            sampler.prefetch_next_update(n_local, stream_extra=2)     # 2 per env: the per-step draws come partly from the stream
        store.reconcile_seeds(set(int(x) for x in sampler.seeds if x >= 0))
        log.append((bool(replay), [int(s) for s in seeds], float(sampler.solvable_mass), float(sampler.max_score)))
    return log


def _state(s):
    return {k: np.array(getattr(s, k)) for k in ("seeds", "seed_scores", "seed_staleness", "unseen_seed_weights",
                                                 "partial_seed_scores", "partial_seed_steps", "seed_solvable")}


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from level_replay.sharded import ShardedLevelSampler, ShardedLevelStore
    np.random.seed(7 if rank == 0 else 12345)                        # only rank 0's stream may matter
    sampler = ShardedLevelSampler(_plr_args(B_LOCAL), group=dist.group.WORLD)
    store = ShardedLevelStore(group=dist.group.WORLD)
    log = _drive(sampler, store, rank * B_LOCAL, B_LOCAL, lambda recs: sampler.update_with_records(recs, device="cpu"), prefetch=True)
    if rank == 0:
        assert getattr(sampler.sampler, "prefetch_hits", 0) >= UPDATES - 2           # the speculation was consumed, not dropped
        assert getattr(sampler.sampler, "stream_hits", 0) >= 3
    with open(os.path.join(out_dir, f"r{rank}.pkl"), "wb") as f:
        pickle.dump({"log": log, "state": _state(sampler.sampler) if rank == 0 else None,
                     "store": (store.seed2level, store.seed2parent, store.next_seed)}, f)
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("WORLD", [2, 8])
def test_sharded_level_replay_equals_single_process(WORLD):
    from level_replay import LevelSampler, LevelStore
    np.random.seed(7)
    ref_sampler, ref_store = LevelSampler(**_plr_args(B_LOCAL * WORLD)), LevelStore()
    ref_log = _drive(ref_sampler, ref_store, 0, B_LOCAL * WORLD, ref_sampler.update_with_records)
    assert any(r[0] for r in ref_log) and not all(r[0] for r in ref_log)       # both branches exercised
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(WORLD, 29541 + WORLD, d), nprocs=WORLD, join=True)
        got = [pickle.load(open(os.path.join(d, f"r{r}.pkl"), "rb")) for r in range(WORLD)]
    for u, ref in enumerate(ref_log):
        seeds = [x for g in got for x in g["log"][u][1]]
        assert [g["log"][u][0] for g in got] == [ref[0]] * WORLD, u
        assert seeds == ref[1], u
        for g in got:
            assert g["log"][u][2:] == ref[2:], u
    for k, v in _state(ref_sampler).items():
        assert np.array_equal(got[0]["state"][k], v, equal_nan=True), k
    for g in got:                                                    # replicated store identical on every rank
        assert g["store"] == (ref_store.seed2level, ref_store.seed2parent, ref_store.next_seed)
