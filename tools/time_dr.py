"""Step-kernel time with auto-reset to RANDOM levels (DR without PLR: DCD_STEP_RESET_RANDOM) vs plain auto-reset."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import torch, dcd_hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for flags, name in ((0, "auto-reset to the same level"), (dcd_hip.STEP_RESET_RANDOM, "auto-reset to a random level")):
    eng = dcd_hip.MultiGridEngine(B, n_clutter=25, choose_goal_last=1, max_steps=250, max_episode_steps=250)
    eng.reset_random()
    out = eng.step_buffers(u8=False)
    acts = [dcd_hip.engine.random_actions(B, 0, 3, t, 7, "cuda") for t in range(600)]
    for t in range(40): eng.step(acts[t], flags, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(40, 600): eng.step(acts[t], flags, out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 560
    print(f"{name:34s} {dt * 1e6:8.1f} us/step  {B / dt / 1e6:8.1f} M env-steps/s  errors {eng.poll_errors()}")
