"""Device time of the teacher's constructive episode (reset + 27 step_adversary launches, synthetic actions) replayed from one
HIP graph at B envs, with different output sets: where a teacher step's 6.6 us go."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import torch, dcd_hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda"
eng = dcd_hip.MultiGridEngine(B, device=dev, size=15, agent_view_size=5, max_steps=250, max_episode_steps=250, n_clutter=25,
                              choose_goal_last=1, random_z_dim=50)
z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)
T = 27
full = [dict(image_u8=z(B, 15, 15, 3, dt=torch.uint8), time_step=z(B, 1), random_z=z(B, 50), done=z(B, dt=torch.uint8)) for _ in range(T + 1)]
variants = {
    "u8 image + time_step + z + done": full,
    "no image": [{k: v for k, v in o.items() if k != "image_u8"} for o in full],
    "no z": [{k: v for k, v in o.items() if k != "random_z"} for o in full],
    "done only": [{"done": o["done"]} for o in full],
    "fp32 image": [dict(o, image_f32=z(B, 3, 15, 15)) for o in [{k: v for k, v in o.items() if k != "image_u8"} for o in full]],
}
loc = dcd_hip.engine.random_actions(B, 0, 2, 0, eng.n_adv_actions, dev, T=T)
for name, outs in variants.items():
    def episode():
        eng.reset(outs[0])
        for k in range(T):
            eng.step_adversary(loc[k], outs[k + 1])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        episode(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            episode()
        for _ in range(5): g.replay()
        s.synchronize(); t = time.perf_counter()
        n = 200
        for _ in range(n): g.replay()
        s.synchronize()
    print(f"{name:36s} {(time.perf_counter() - t) / n / (T + 1) * 1e6:6.2f} us per launch", flush=True)
print("errors", eng.poll_errors())
