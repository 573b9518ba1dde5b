"""Per-op device time of ONE student PPO update (5 epochs, B = 4096, T = 256) grouped by input shape: which GEMM / reduction / copy
is which layer.  Output: wall time, summed device time, the table (80 rows)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from tests.test_gpu_runner import make_args, build
args = make_args(ued_algo="paired", num_processes=4096, num_steps=256, ppo_epoch=5, adv_ppo_epoch=5)
runner, venv = build(args)
runner.run(); runner.run()
names = sys.argv[1:] or ["agent"]
for name in names:
    a = runner.agents[name]
    a.update(); torch.cuda.synchronize()
    walls = []
    for _ in range(3):
        t = time.perf_counter(); a.update(); torch.cuda.synchronize(); walls.append((time.perf_counter() - t) * 1e3)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        a.update(); torch.cuda.synchronize()
    dev_us = sum(k.self_device_time_total for k in prof.key_averages())
    print(f"{name}: update wall {min(walls):.1f} ms (3 runs: {[round(w, 1) for w in walls]}); summed device time {dev_us/1e3:.1f} ms", flush=True)
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_device_time_total", row_limit=80, max_name_column_width=48, max_shapes_column_width=70))
