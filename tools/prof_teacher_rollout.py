import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from arguments import parser
from envs.runners import AdversarialRunner
from util import create_parallel_env, make_agent, seed
args = parser.parse_args([])
for k, v in dict(env_name="MultiGrid-GoalLastFewerBlocksAdversarial-v0", num_processes=4096, num_steps=256, ued_algo="paired",
                 handle_timelimits=True, recurrent_adversary_env=True).items():
    setattr(args, k, v)
seed(1); dev = "cuda:0"
venv, ued = create_parallel_env(args, device=dev)
agent = make_agent("agent", venv, args, dev); ant = make_agent("adversary_agent", venv, args, dev); teacher = make_agent("adversary_env", venv, args, dev)
runner = AdversarialRunner(args=args, venv=venv, agent=agent, ued_venv=ued, adversary_agent=ant, adversary_env=teacher, train=True, device=dev)
for _ in range(3): runner._teacher_rollout(teacher, 27)
torch.cuda.synchronize(); t = time.perf_counter(); runner._teacher_rollout(teacher, 27); torch.cuda.synchronize(); print("ms", (time.perf_counter() - t) * 1e3)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    runner._teacher_rollout(teacher, 27); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=16, max_name_column_width=50))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12, max_name_column_width=50))
