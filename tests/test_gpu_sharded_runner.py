"""-m gpu: the training driver's sharded path end to end — 2 ranks x 32 envs (both on the test box's single GPU,
gloo as the transport because RCCL refuses two ranks on one device), robust PLR and ACCEL: gradient all-reduce
in PPO, episode-score records all-gathered into rank 0's LevelSampler, replicated LevelStore, rank-0 decisions.
Checks that the ranks stay in lock-step (same replay decisions, same parameters after every update — they see
the same all-reduced gradients) and that the replay buffer fills from BOTH ranks' levels."""
import os
import pickle
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, out_dir, accel):
    for p in (ROOT, os.path.join(ROOT, "dcd-isaac_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.test_gpu_runner import make_args
    from envs.runners import AdversarialRunner
    from util import create_parallel_env, make_agent, make_plr_args, seed
    from util import distributed as D
    kw = dict(ued_algo="domain_randomization", use_plr=True, use_reset_random_dr=True, no_exploratory_grad_updates=True,
              level_replay_strategy="positive_value_loss", level_replay_score_transform="rank", level_replay_prob=0.5,
              level_replay_rho=0.2, level_replay_seed_buffer_size=100, staleness_coef=0.3, log_plr_buffer_stats=True,
              recurrent_adversary_env=False, num_steps=64, num_processes=32, env_name="MultiGrid-MiniGoalLastAdversarial-v0")
    if accel:
        kw.update(use_editor=True, level_editor_prob=1.0, num_edits=3, base_levels="batch")
    args = make_args(**kw)
    dev = "cuda:0"
    seed(args.seed + rank)
    pg = dist.group.WORLD
    venv, ued = create_parallel_env(args, device=dev, env_index_base=rank * args.num_processes)
    # ranks seed differently (args.seed + rank): PPO broadcasts rank 0's initial weights
    agent = make_agent("agent", venv, args, dev, process_group=pg)
    teacher = make_agent("adversary_env", venv, args, dev, process_group=pg)
    runner = AdversarialRunner(args=args, venv=venv, agent=agent, ued_venv=ued, adversary_env=teacher, train=True,
                               plr_args=make_plr_args(args, venv.observation_space, venv.action_space), device=dev,
                               process_group=pg)
    log = []
    for j in range(6):
        stats = D.reduce_stats(runner.run())
        w = torch.cat([p.detach().flatten() for p in agent.algo.actor_critic.parameters()]).double()
        log.append({"replay": bool(runner.sampled_level_info["level_replay"]), "wsum": float(w.sum()), "wabs": float(w.abs().sum()),
                    "steps": stats["steps"], "seeds": list(runner.current_level_seeds), "n_levels": len(runner.level_store),
                    "mass": stats.get("agent_plr_passable_mass"), "edits": runner.total_num_edits})
    sampler = runner.level_samplers["agent"].sampler
    with open(os.path.join(out_dir, f"r{rank}.pkl"), "wb") as f:
        pickle.dump({"log": log, "buffer": None if sampler is None else [int(s) for s in sampler.seeds if s >= 0],
                     "store": dict(runner.level_store.seed2level)}, f)
    venv.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("accel", [False, True])
def test_two_rank_robust_plr(accel):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, 29551 + int(accel), d, accel), nprocs=2, join=True)
        r0, r1 = (pickle.load(open(os.path.join(d, f"r{r}.pkl"), "rb")) for r in range(2))
    assert [x["replay"] for x in r0["log"]] == [x["replay"] for x in r1["log"]]
    assert any(x["replay"] for x in r0["log"])
    for a, b in zip(r0["log"], r1["log"]):
        assert a["wsum"] == b["wsum"] and a["wabs"] == b["wabs"]            # same all-reduced gradients -> same weights
        assert a["steps"] == b["steps"] and a["n_levels"] == b["n_levels"] and a["mass"] == b["mass"]
        assert a["edits"] == b["edits"]
    assert r0["log"][-1]["wsum"] != r0["log"][0]["wsum"]                    # and they did learn
    assert r0["log"][-1]["steps"] == 2 * (6 + r0["log"][-1]["edits"]) * 32 * 64
    assert r0["store"] == r1["store"] and r1["buffer"] is None
    seeds0 = {s for x in r0["log"] if not x["replay"] for s in x["seeds"]}
    seeds1 = {s for x in r1["log"] if not x["replay"] for s in x["seeds"]}
    assert set(r0["buffer"]) & seeds0 and set(r0["buffer"]) & seeds1        # buffer holds levels from both ranks
    if accel:
        assert r0["log"][-1]["edits"] > 0
