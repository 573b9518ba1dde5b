"""cProfile of AdversarialRunner.run() (PAIRED, 4096 envs) on the calling thread: where the host time of an update goes."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import torch
from arguments import parser
from envs.runners import AdversarialRunner
from util import create_parallel_env, make_agent, seed
args = parser.parse_args([])
for k, v in dict(env_name="MultiGrid-GoalLastFewerBlocksAdversarial-v0", num_processes=4096, num_steps=256, ued_algo="paired",
                 handle_timelimits=True, recurrent_adversary_env=True, log_action_complexity=True).items():
    setattr(args, k, v)
seed(1); dev = "cuda:0"
venv, ued = create_parallel_env(args, device=dev)
agents = [make_agent(n, venv, args, dev) for n in ("agent", "adversary_agent", "adversary_env")]
runner = AdversarialRunner(args=args, venv=venv, agent=agents[0], ued_venv=ued, adversary_agent=agents[1], adversary_env=agents[2], train=True, device=dev)
for _ in range(3): runner.run()
torch.cuda.synchronize()
pr = cProfile.Profile(); t = time.perf_counter(); pr.enable(); runner.run(); torch.cuda.synchronize(); pr.disable()
print("run() s", time.perf_counter() - t)
pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
