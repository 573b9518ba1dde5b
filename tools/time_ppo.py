"""Op-level breakdown (torch.profiler, device time) of one student and one teacher PPO update at B=4096."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from arguments import parser
from envs.runners import AdversarialRunner
from util import create_parallel_env, make_agent, seed

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
args = parser.parse_args([])
for k, v in dict(env_name="MultiGrid-GoalLastFewerBlocksAdversarial-v0", num_processes=B, num_steps=256, ued_algo="paired",
                 handle_timelimits=True, recurrent_adversary_env=True, ppo_epoch=1, adv_ppo_epoch=1).items():
    setattr(args, k, v)
seed(1)
dev = "cuda:0"
venv, ued = create_parallel_env(args, device=dev)
agent = make_agent("agent", venv, args, dev); ant = make_agent("adversary_agent", venv, args, dev); teacher = make_agent("adversary_env", venv, args, dev)
runner = AdversarialRunner(args=args, venv=venv, agent=agent, ued_venv=ued, adversary_agent=ant, adversary_env=teacher, train=True, device=dev)
runner.run()
for name, a in (("student", agent), ("teacher", teacher)):
    a.update(); torch.cuda.synchronize()
    t = time.perf_counter(); a.update(); torch.cuda.synchronize(); print(name, "one-epoch update ms", (time.perf_counter() - t) * 1e3)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        a.update(); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=22, max_name_column_width=60))
