"""Teacher rollout only (27 steps, B = 4096): for a kernel-level profile under rocprofv3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import torch
from arguments import parser
from envs.runners import AdversarialRunner
from util import create_parallel_env, make_agent, seed
args = parser.parse_args([])
for k, v in dict(env_name="MultiGrid-GoalLastFewerBlocksAdversarial-v0", num_processes=4096, num_steps=256, ued_algo="paired",
                 handle_timelimits=True, recurrent_adversary_env=True).items():
    setattr(args, k, v)
seed(1)
venv, ued = create_parallel_env(args, device="cuda:0")
agent = make_agent("agent", venv, args, "cuda:0"); teacher = make_agent("adversary_env", venv, args, "cuda:0")
runner = AdversarialRunner(args=args, venv=venv, agent=agent, ued_venv=ued, adversary_env=teacher, train=True, device="cuda:0")
for _ in range(3): runner._teacher_rollout(teacher, 27)          # eager warm-up, capture, first replay
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): runner._teacher_rollout(teacher, 27)
torch.cuda.synchronize(); print("teacher rollout ms", (time.perf_counter() - t) / 10 * 1e3)
