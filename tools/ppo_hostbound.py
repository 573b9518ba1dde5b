"""Is the PPO update host-bound or GPU-bound?  One student update (5 epochs, B = 4096, T = 256) alone on the device: wall time
vs the sum of its kernels' device time (torch.profiler), and the number of kernel launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from tests.test_gpu_runner import make_args, build
os.environ["DCD_CONCURRENT_PPO"] = "0"
args = make_args(ued_algo="paired", num_processes=4096, num_steps=256, ppo_epoch=5, adv_ppo_epoch=5)
runner, venv = build(args)
runner.run(); runner.run()
for name in ("agent", "adversary_env"):
    a = runner.agents[name]
    a.update(); torch.cuda.synchronize()
    t = time.perf_counter(); a.update(); torch.cuda.synchronize(); wall = time.perf_counter() - t
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        a.update(); torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    dev_us = sum(e.device_time for e in prof.key_averages() for _ in [0] if False) or sum(k.self_device_time_total for k in prof.key_averages())
    n_launch = sum(k.count for k in prof.key_averages() if k.self_device_time_total > 0)
    print(f"{name}: update wall {wall*1e3:.1f} ms; summed device time {dev_us/1e3:.1f} ms; ops with device time {n_launch}", flush=True)
    print(prof.key_averages().table(sort_by="self_device_time_total", row_limit=28, max_name_column_width=70))
