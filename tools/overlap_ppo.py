"""Do the two students' PPO updates overlap when issued from two host threads on two streams?  (They are independent:
different nets, different storages.)  Prints sequential vs concurrent wall time and checks the weights are identical."""
import copy, os, sys, threading, time
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
agents = {n: make_agent(n, venv, args, dev) for n in ("agent", "adversary_agent", "adversary_env")}
runner = AdversarialRunner(args=args, venv=venv, agent=agents["agent"], ued_venv=ued, adversary_agent=agents["adversary_agent"],
                           adversary_env=agents["adversary_env"], train=True, device=dev)
runner.run()
torch.cuda.synchronize()
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["agent", "adversary_agent"]
group = [agents[n] for n in names]


def snapshot():
    return [copy.deepcopy(a.algo.actor_critic.state_dict()) for a in group], [copy.deepcopy(a.algo.optimizer.state_dict()) for a in group]


def restore(snap):
    for a, sd, od in zip(group, *snap):
        a.algo.actor_critic.load_state_dict(sd); a.algo.optimizer.load_state_dict(od)


def sequential():
    for a in group:
        a.update()


streams = [torch.cuda.Stream() for _ in group]


def concurrent():
    cur = torch.cuda.current_stream()
    def work(a, s):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            a.update()
    th = [threading.Thread(target=work, args=(a, s)) for a, s in zip(group, streams)]
    for t in th: t.start()
    for t in th: t.join()
    for s in streams: cur.wait_stream(s)


snap = snapshot()
res = {}
runs = []
for name, fn in (("sequential", sequential), ("concurrent", concurrent), ("sequential", sequential), ("concurrent", concurrent), ("concurrent", concurrent)):
    restore(snap); torch.cuda.synchronize()
    t = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    res[name] = [{k: v.clone() for k, v in a.algo.actor_critic.state_dict().items()} for a in group]
    runs.append((name, res[name]))
    print(f"{name:11s} {dt * 1e3:8.1f} ms", flush=True)
for i in range(len(runs)):
    for j in range(i + 1, len(runs)):
        d = max(float((x[k].float() - y[k].float()).abs().max()) for x, y in zip(runs[i][1], runs[j][1]) for k in x)
        print(f"run {i} ({runs[i][0]}) vs run {j} ({runs[j][0]}): max |dw| = {d:.3e}")
same = all(torch.equal(x[k], y[k]) for x, y in zip(res["sequential"], res["concurrent"]) for k in x)
print("weights identical:", same)
