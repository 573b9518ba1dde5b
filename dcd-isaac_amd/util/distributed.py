"""One process per GPU over torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" on CPU tests).

The env batch is sharded by rank (global env g -> rank g // B): rollouts, GAE and episode
scoring need no communication.  The exchange steps of the path are:
  * gradients / advantage moments / loss means inside PPO (algos/ppo.py);
  * per-episode score records, per-env regrets and new-level encodings gathered to rank 0's single
    LevelSampler / LevelStore; replay decisions, sampled seeds and their encodings broadcast back.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init(backend=None, device=None):
    """Initialise from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run); no-op for one process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # DCD_DIST_BACKEND=gloo: rehearsal with several ranks on ONE GPU (RCCL refuses two ranks per device)
    backend = backend or os.environ.get("DCD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, **kw)
    return world


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def all_gather_records(recs, device, group=None):
    """recs: numpy structured array (8 x 4-byte fields) of this rank -> list of per-rank arrays on every
    rank.  Variable lengths: counts first, then one padded all_gather."""
    dtype = recs.dtype
    if world_size() == 1:
        return [recs]
    n = torch.tensor([len(recs)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world_size())]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    buf = torch.zeros((mx, 8), dtype=torch.int32, device=device)
    if len(recs):
        buf[:len(recs)] = torch.from_numpy(np.ascontiguousarray(recs).view(np.int32).reshape(-1, 8)).to(device)
    out = [torch.zeros_like(buf) for _ in range(world_size())]
    dist.all_gather(out, buf, group=group)
    return [o[:c].cpu().numpy().view(dtype).reshape(c) for o, c in zip(out, counts)]


def all_gather_tensor(t):
    """[B, ...] per rank -> [world*B, ...] on every rank (rank-major == global env order)."""
    if world_size() == 1:
        return t
    out = [torch.zeros_like(t) for _ in range(world_size())]
    dist.all_gather(out, t.contiguous())
    return torch.cat(out, dim=0)


def broadcast_object(obj, src=0):
    if world_size() == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def scatter_rows(t, rows_per_rank):
    """Rank 0 holds [world*rows, ...]; every rank returns its own [rows, ...] slice (broadcast + slice)."""
    if world_size() == 1:
        return t
    dist.broadcast(t, src=0)
    r = rank()
    return t[r * rows_per_rank:(r + 1) * rows_per_rank]


SUM_KEYS = ("steps", "total_episodes", "total_seeds")


def reduce_stats(stats, sum_keys=SUM_KEYS):
    """Per-rank `AdversarialRunner.run()` stats -> job-wide stats on every rank: counters are summed, every
    other numeric entry is averaged over the ranks that report it (None = not reported, as in the reference's
    plr_* / non-plr_* key pairs); non-numeric entries keep rank 0's value."""
    if world_size() == 1:
        return stats
    parts = [None] * world_size()
    num = {k: float(v) for k, v in stats.items() if isinstance(v, (int, float, np.floating, np.integer)) and not isinstance(v, bool)}
    dist.all_gather_object(parts, num)
    out = dict(stats)
    for k in num:
        vals = [p[k] for p in parts if k in p]
        out[k] = float(np.sum(vals)) if k in sum_keys else float(np.mean(vals))
    return out
