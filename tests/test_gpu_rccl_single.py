"""-m gpu: the multi-rank code path through RCCL itself, on the one GPU a test box has.

Every other multi-rank test uses gloo (RCCL refuses two ranks on one device).  A world_size-1 NCCL (= RCCL) group is legal on one
GPU and executes every collective the sharded training job issues: the broadcast of rank 0's initial weights and the bucketed
`all_reduce(async_op=True)` launched from the post-accumulate-grad hooks (algos/ppo.py), the advantage-moment all-reduce, the
fixed-capacity `all_gather_into_tensor` of the episode-score records from the scorer's device buffers
(util/distributed.py::gather_records_device), the int64 tensor collectives that carry replay decisions / seeds / new levels
(level_replay/sharded.py).  With one rank every one of them is the identity, so the run must equal the run without a group
BIT FOR BIT — weights, replay decisions, seeds, logged statistics.  The child process is fresh (spawn) and creates the group
before its first GPU call."""
import os
import pickle
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _child(_, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "dcd-isaac_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    calls = {}

    def counted(name):
        orig = getattr(dist, name)

        def f(*a, **k):
            t = a[1] if name in ("all_gather_into_tensor", "all_gather") else a[0]
            on_gpu = bool(getattr(t, "is_cuda", False))
            key = name + ("[async]" if k.get("async_op") else "") + ("" if on_gpu else "[host]")
            calls[key] = calls.get(key, 0) + 1
            return orig(*a, **k)
        setattr(dist, name, f)
    for n in ("all_reduce", "all_gather_into_tensor", "broadcast", "all_gather"):
        counted(n)

    from tests.test_gpu_runner import make_args
    from envs.runners import AdversarialRunner
    from util import create_parallel_env, make_agent, make_plr_args, seed

    def run(kind, group):
        if kind == "robust_plr":
            kw = dict(ued_algo="domain_randomization", use_plr=True, use_reset_random_dr=True, no_exploratory_grad_updates=True,
                      level_replay_strategy="positive_value_loss", level_replay_score_transform="rank", level_replay_prob=0.8,
                      level_replay_rho=0.1, level_replay_seed_buffer_size=100, staleness_coef=0.3, recurrent_adversary_env=False,
                      num_steps=64, num_processes=64, env_name="MultiGrid-MiniGoalLastAdversarial-v0")
            n_updates = 4
        else:
            kw = dict(ued_algo="paired", num_steps=48, num_processes=64)
            n_updates = 2
        args = make_args(**kw)
        seed(args.seed)
        venv, ued = create_parallel_env(args, device="cuda:0")
        gk = {} if group is None else {"process_group": group}
        agent = make_agent("agent", venv, args, "cuda:0", **gk)
        adv_agent = make_agent("adversary_agent", venv, args, "cuda:0", **gk) if kind == "paired" else None
        teacher = make_agent("adversary_env", venv, args, "cuda:0", **gk)
        plr = make_plr_args(args, venv.observation_space, venv.action_space) if args.use_plr else None
        runner = AdversarialRunner(args=args, venv=venv, agent=agent, ued_venv=ued, adversary_agent=adv_agent, adversary_env=teacher,
                                   train=True, plr_args=plr, device="cuda:0", **gk)
        log = []
        for _ in range(n_updates):
            stats = runner.run()
            nets = [agent, teacher] + ([adv_agent] if adv_agent is not None else [])
            w = [torch.cat([p.detach().flatten() for p in a.algo.actor_critic.parameters()]).double() for a in nets]
            log.append({"replay": bool(runner.sampled_level_info["level_replay"]) if runner.sampled_level_info else None,
                        "w": [(float(x.sum()), float(x.abs().sum())) for x in w], "seeds": [int(s) for s in (runner.current_level_seeds or [])],
                        "stats": {k: float(v) for k, v in stats.items() if isinstance(v, (int, float)) and k not in ("sps",)}})
        venv.close()
        return log

    out = {}
    for kind in ("robust_plr", "paired"):
        calls.clear()
        with_group = run(kind, dist.group.WORLD)
        out[kind] = {"calls": dict(calls), "group": with_group}
        calls.clear()
        out[kind]["plain"] = run(kind, None)
        out[kind]["calls_plain"] = dict(calls)
    with open(out_path, "wb") as f:
        pickle.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_rccl_group_runs_every_collective_and_equals_the_plain_run():
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "out.pkl")
        mp.spawn(_child, args=(29571, path), nprocs=1, join=True)
        out = pickle.load(open(path, "rb"))
    for kind in ("robust_plr", "paired"):
        o = out[kind]
        c = o["calls"]
        assert o["calls_plain"] == {}, o["calls_plain"]                       # no group, no collective
        assert c.get("broadcast", 0) >= 2                                     # initial weights of every agent, on the GPU
        assert c.get("all_reduce[async]", 0) >= 1, c                          # gradient buckets from the backward hooks
        assert c.get("all_reduce", 0) >= 1, c                                 # advantage moments
        if kind == "robust_plr":
            assert c.get("all_gather_into_tensor", 0) >= 1, c                 # score records straight from the device buffers
            assert any(x["replay"] for x in o["group"]) and not all(x["replay"] for x in o["group"])
        assert not any(k.endswith("[host]") for k in c), c                    # RCCL: every payload lives on the GPU
        for a, b in zip(o["group"], o["plain"]):
            assert a["replay"] == b["replay"] and a["seeds"] == b["seeds"]
            assert a["w"] == b["w"]                                           # bit-equal parameters after every update
            for k in a["stats"]:
                if k in b["stats"] and k != "steps":
                    assert a["stats"][k] == b["stats"][k] or (a["stats"][k] != a["stats"][k] and b["stats"][k] != b["stats"][k]), k
