"""Robust PLR (BASELINE config 3 shape: mg_25b_robust_plr.json) at B = 4096, T = 256 on one GPU: seconds per update,
replay vs exploratory, with the host-side phases of the level sampler broken out."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from arguments import parser
from envs.runners import AdversarialRunner
from util import create_parallel_env, make_agent, make_plr_args, seed

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
flags = json.load(open(os.path.join(ROOT, "tests", "golden", "mg_train_configs.json")))["25_blocks/mg_25b_robust_plr.json"]
args = parser.parse_args([f"--{k}={v}" for k, v in flags.items()])
args.num_processes = B
args.level_replay_rho = 0.2                    # start replaying once 800 of the 4000 buffer slots are filled
seed(1)
dev = "cuda:0"
venv, ued = create_parallel_env(args, device=dev)
agent = make_agent("agent", venv, args, dev); teacher = make_agent("adversary_env", venv, args, dev); teacher.random()
runner = AdversarialRunner(args=args, venv=venv, agent=agent, ued_venv=ued, adversary_env=teacher, train=True,
                           plr_args=make_plr_args(args, venv.observation_space, venv.action_space), device=dev)
SIM = int(os.environ.get("SIM_WORLD", "1"))       # rank 0 of an N-rank job draws N x as many levels: time the host side as if N ranks asked
if SIM > 1:
    ls = runner.level_samplers["agent"]
    draw, pre = ls.sample_replay_levels, ls.prefetch_next_update
    ls.sample_replay_levels = lambda n, **k: draw(n * SIM, **k)[:n]
    ls.prefetch_next_update = lambda n, stream_extra=None: pre(n * SIM, stream_extra=(stream_extra or 0) * SIM)
import cProfile, pstats
prof_replay = prof_explore = False
for j in range(12):
    torch.cuda.synchronize(); t = time.perf_counter()
    pr = cProfile.Profile() if (j >= 4 and not os.environ.get("NO_PROFILE")) else None
    if pr: pr.enable()
    stats = runner.run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    if pr:
        pr.disable()
        rep = bool(runner.sampled_level_info["level_replay"])
        if (rep and not prof_replay) or (not rep and not prof_explore):
            print("---- profile of a", "replay" if rep else "exploratory", "update")
            pstats.Stats(pr).sort_stats("tottime").print_stats(14)
            prof_replay, prof_explore = prof_replay or rep, prof_explore or not rep
    print(json.dumps({"update": j, "replay": bool(runner.sampled_level_info["level_replay"]), "seconds": round(dt, 3),
                      "env_steps_per_s": round(B * 256 / dt), "buffer": int(runner.level_samplers["agent"].working_seed_buffer_size), "prefetch_hits": getattr(runner.level_samplers["agent"], "prefetch_hits", 0), "stream_hits": getattr(runner.level_samplers["agent"], "stream_hits", 0), "sim_world": SIM}), flush=True)
