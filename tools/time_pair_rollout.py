"""Times AdversarialRunner._paired_student_rollouts (PAIRED, B=4096, T=256; both students in lock-step, policy in the loop,
one HIP graph) — the S_roll leg of bench.py — and its graph alone.  DCD_PAIR_POLICY=0/1 for a same-box A/B."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import torch
from arguments import parser
from envs.runners import AdversarialRunner
from util import create_parallel_env, make_agent, seed

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
args = parser.parse_args([])
for k, v in dict(env_name="MultiGrid-GoalLastFewerBlocksAdversarial-v0", num_processes=B, num_steps=256, ued_algo="paired",
                 handle_timelimits=True, recurrent_adversary_env=True).items():
    setattr(args, k, v)
seed(1)
dev = "cuda:0"
venv, ued = create_parallel_env(args, device=dev)
agents = [make_agent(n, venv, args, dev) for n in ("agent", "adversary_agent", "adversary_env")]
runner = AdversarialRunner(args=args, venv=venv, agent=agents[0], ued_venv=ued, adversary_agent=agents[1], adversary_env=agents[2],
                           train=True, device=dev)
runner._teacher_rollout(agents[2], 27)
pair = (agents[0], agents[1])
for _ in range(3):
    runner._paired_student_rollouts(pair, 256, track_seeds=False)
torch.cuda.synchronize()
n = 8
t = time.perf_counter()
for _ in range(n):
    runner._paired_student_rollouts(pair, 256, track_seeds=False)
torch.cuda.synchronize()
whole = (time.perf_counter() - t) / n
graphs = [g for k, g in runner._graphs.items() if k[0] == "pair"][0]
t = time.perf_counter()
for _ in range(n):
    for g in graphs:
        g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t) / n
print(json.dumps({"pair_policy": os.environ.get("DCD_PAIR_POLICY", "1"), "both_rollouts_ms": whole * 1e3, "graph_only_ms": graph * 1e3,
                  "S_roll_env_steps_per_s_graph": 2 * B * 256 / graph, "S_roll_whole": 2 * B * 256 / whole}))
