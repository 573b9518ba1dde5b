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


def comm_device(group=None, device=None):
    """Where collective payloads must live: the GPU for RCCL ("nccl"), host memory for gloo."""
    if dist.get_backend(group) == "nccl":
        return torch.device(device) if device is not None and torch.device(device).type == "cuda" else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_records_device(recs_dev, n_dev, group=None, cap=None):
    """Per-episode score records of every rank, gathered WITHOUT a count exchange and without a host hop on the send
    side: ONE fixed-capacity all-gather of int32 words [count | recs[:cap]] straight from the scorer's
    device buffers (RCCL over xGMI on the GPU path; the same call on CPU tensors under gloo).  recs_dev: int32
    [max_recs, 8] device tensor, n_dev: int32 [1] device tensor (the scorer's record count, never read on the host
    here).  Returns (counts list, gathered int32 tensor [world, 1 + cap*8]) — the tensor stays where the collective
    put it; only the consumer (rank 0) copies its valid rows to the host.  cap: records per rank carried by the first
    exchange (default 4 per env is what a 256-step rollout of 250-step episodes needs about twice over); when some
    rank has more, every rank sees it in the gathered counts and a second exchange at the largest count follows."""
    world = dist.get_world_size(group)
    dev = comm_device(group, recs_dev.device)
    max_recs = recs_dev.shape[0]
    cap = min(max_recs, cap if cap is not None else max_recs)
    while True:
        send = torch.empty((1 + cap * 8,), dtype=torch.int32, device=recs_dev.device)
        send[:1].copy_(n_dev.view(-1)[:1])
        send[1:].copy_(recs_dev[:cap].reshape(-1))
        send = send.to(dev)
        recv = torch.empty((world, 1 + cap * 8), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(recv.view(-1), send, group=group)
        counts = recv[:, 0].cpu().tolist()                        # 4 bytes per rank: the only host read every rank makes
        if max(counts) <= cap:
            return counts, recv
        cap = min(max_recs, max(counts))                          # same decision on every rank: the collective order holds


def all_gather_records(recs, device, group=None):
    """Host records (numpy structured array, 8 x 4-byte fields) of this rank -> list of per-rank arrays on every rank
    (used where the records are produced on the host: CPU tests, the oracle)."""
    dtype = recs.dtype
    if world_size() == 1:
        return [recs]
    dev = comm_device(group, device)
    cap = max(max(c[0] for c in gather_int_lists([len(recs)], group=group, device=device)), 1)     # host records: lengths first
    buf = torch.zeros((cap, 8), dtype=torch.int32)
    if len(recs):
        buf[:len(recs)] = torch.from_numpy(np.ascontiguousarray(recs).view(np.int32).reshape(-1, 8))
    counts, recv = gather_records_device(buf.to(dev), torch.tensor([len(recs)], dtype=torch.int32, device=dev), group=group, cap=cap)
    body = recv[:, 1:].cpu().numpy()
    return [body[r, :c * 8].copy().view(dtype).reshape(c) for r, c in enumerate(counts)]


def gather_int_lists(values, group=None, device=None):
    """Every rank contributes a list of ints (ragged); every rank gets the list of lists.  Two tensor collectives
    (lengths, padded payload) instead of a pickled all_gather_object."""
    world = dist.get_world_size(group)
    dev = comm_device(group, device)
    n = torch.tensor([len(values)], dtype=torch.int64, device=dev)
    lens = torch.empty((world,), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(lens, n, group=group)
    lens = lens.cpu().tolist()
    mx = max(max(lens), 1)
    send = torch.zeros((mx,), dtype=torch.int64)
    if len(values):
        send[:len(values)] = torch.as_tensor(list(values), dtype=torch.int64)
    recv = torch.empty((world, mx), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(recv.view(-1), send.to(dev), group=group)
    recv = recv.cpu()
    return [recv[r, :c].tolist() for r, c in enumerate(lens)]


def broadcast_ints(values, n, group=None, device=None):
    """Rank 0's list of `n` ints to every rank (n known everywhere): one tensor broadcast."""
    dev = comm_device(group, device)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    t = torch.zeros((max(int(n), 1),), dtype=torch.int64)
    if dist.get_rank(group) == 0 and n:
        t[:n] = torch.as_tensor(list(values), dtype=torch.int64)
    t = t.to(dev)
    dist.broadcast(t, src=src, group=group)
    return t[:n].cpu().tolist()


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
