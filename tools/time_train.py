"""Times AdversarialRunner.run() (PAIRED, B=4096, T=256) phases on one GPU: S_roll / S_train of SURVEY §8d."""
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
                 handle_timelimits=True, recurrent_adversary_env=True, ppo_epoch=5, adv_ppo_epoch=5).items():
    setattr(args, k, v)
seed(1)
dev = "cuda:0"
venv, ued = create_parallel_env(args, device=dev)
agent = make_agent("agent", venv, args, dev); ant = make_agent("adversary_agent", venv, args, dev); teacher = make_agent("adversary_env", venv, args, dev)
runner = AdversarialRunner(args=args, venv=venv, agent=agent, ued_venv=ued, adversary_agent=ant, adversary_env=teacher, train=True, device=dev)
def sync(): torch.cuda.synchronize()
# rollout only (no update): S_roll
runner.eval(); runner.is_training = False
runner._teacher_rollout(teacher, 27)
sync(); t = time.perf_counter()
for _ in range(3): runner._student_rollout(agent, 256, False, None, {})
sync(); print('3 warm-up rollouts (incl. graph capture if enabled) s:', time.perf_counter() - t)
sync(); t = time.perf_counter()
for _ in range(2): runner._student_rollout(agent, 256, False, None, {})
sync(); dt = (time.perf_counter() - t) / 2
print(json.dumps({"S_roll_steps_per_s": B * 256 / dt, "student_rollout_ms": dt * 1e3}))
if os.environ.get("ROLLOUT_ONLY"): sys.exit(0)
sync(); t = time.perf_counter(); runner._teacher_rollout(teacher, 27); sync(); print("teacher rollout ms", (time.perf_counter() - t) * 1e3)
runner.train(); runner.is_training = True
for j in range(int(os.environ.get("N_UPDATES", "2"))):
    sync(); t = time.perf_counter(); stats = runner.run(); sync(); dt = time.perf_counter() - t
    print(json.dumps({"update": j, "run_s": dt, "S_train_sps_reference_formula": B * 256 / dt, "true_steps_per_s": (2 * B * 256 + B * 27) / dt,
                      "agent_value_loss": stats["agent_value_loss"], "mean_env_return": stats["mean_env_return"]}))
# ---- phase timing of one more update
import torch.autograd.profiler as prof
def timed(name, fn):
    sync(); t = time.perf_counter(); r = fn(); sync(); print(f"{name:34s} {(time.perf_counter()-t)*1e3:10.1f} ms"); return r
timed("teacher rollout", lambda: runner._teacher_rollout(teacher, 27))
timed("student rollout", lambda: runner._student_rollout(agent, 256, False, None, {}))
st = agent.storage
nv = agent.get_value(st.get_obs(-1), st.get_recurrent_hidden_state(-1), st.masks[-1]).detach()
timed("compute_returns (trunc V + GAE)", lambda: st.compute_returns(nv, True, 0.995, 0.95))
timed("return stats", lambda: runner._rollout_return_stats(st))
timed("student PPO update (5 epochs)", lambda: agent.update())
ts = teacher.storage
nv = teacher.get_value(ts.get_obs(-1), ts.get_recurrent_hidden_state(-1), ts.masks[-1]).detach()
ts.compute_returns(nv, True, 0.995, 0.95)
timed("teacher PPO update (5 epochs)", lambda: teacher.update())
