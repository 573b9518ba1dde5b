"""Per-call time of the level-kernel operations (reset, step_adversary, reset_agent) at B envs, eager launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import torch, dcd_hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda"
eng = dcd_hip.MultiGridEngine(B, device=dev, size=15, agent_view_size=5, max_steps=250, max_episode_steps=250, n_clutter=25,
                              choose_goal_last=1, random_z_dim=50)
z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)
adv = dict(image_f32=z(B, 3, 15, 15), time_step=z(B, 1), random_z=z(B, 50), done=z(B, dt=torch.uint8))
stu = dict(image_f32=z(B, 3, 5, 5), dir_f32=z(B, 1))
loc = dcd_hip.engine.random_actions(B, 0, 2, 0, eng.n_adv_actions, dev, T=27)
def timed(name, fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{name:16s} {(time.perf_counter() - t) / n * 1e6:7.2f} us/call", flush=True)
eng.reset(adv)
for k in range(27): eng.step_adversary(loc[k], adv)
timed("reset_agent", lambda: eng.reset_agent(stu))
timed("step_adversary", lambda: eng.step_adversary(loc[3], adv))
timed("reset", lambda: eng.reset(adv))
timed("get_state", lambda: eng.get_state())
print("errors", eng.poll_errors())
