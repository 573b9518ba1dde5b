"""Is one PPO update run-to-run deterministic?  Same rollout, same initial weights, same minibatch permutations, the update
run twice: max |delta| over the resulting weights, for the student and the teacher, with the switches that route the
update through different kernels.  Usage: python tools/ppo_determinism.py [B] [T]"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import torch
from tests.test_gpu_runner import make_args, build

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64


def probe(label, env):
    for k, v in env.items():
        os.environ[k] = v
    os.environ["DCD_CONCURRENT_PPO"] = "0"
    args = make_args(ued_algo="paired", num_processes=B, num_steps=T, ppo_epoch=2, adv_ppo_epoch=2)
    runner, venv = build(args)
    runner.run()                                   # fills the storages (and performs one update)
    for name in ("agent", "adversary_env"):
        ag = runner.agents[name]
        st = ag.storage
        w0 = copy.deepcopy(ag.algo.actor_critic.state_dict())
        o0 = copy.deepcopy(ag.algo.optimizer.state_dict())
        snap = {k: v.clone() for k, v in vars(st).items() if isinstance(v, torch.Tensor)}
        obs = {k: v.clone() for k, v in st.obs.items()}
        res = []
        for rep in range(3):
            ag.algo.actor_critic.load_state_dict(w0)
            ag.algo.optimizer.load_state_dict(copy.deepcopy(o0))
            for k, v in snap.items():
                getattr(st, k).copy_(v)
            for k, v in obs.items():
                st.obs[k].copy_(v)
            torch.manual_seed(5)
            losses = ag.update()[:3]
            res.append((losses, {k: v.clone() for k, v in ag.algo.actor_critic.state_dict().items()}))
        d = max(float((res[0][1][k] - res[r][1][k]).abs().max()) for r in (1, 2) for k in res[0][1])
        worst = max(((float((res[0][1][k] - res[1][1][k]).abs().max()), k) for k in res[0][1]))
        print(f"{label:34s} {name:14s} max |dw| over 3 repeats {d:.3e} (worst {worst[1]}: {worst[0]:.3e}); losses {[tuple(round(x, 7) for x in r[0]) for r in res]}", flush=True)
    venv.close()


probe("default", {})
probe("DCD_FUSED_LSTM=0", {"DCD_FUSED_LSTM": "0"})
os.environ["DCD_FUSED_LSTM"] = "1"
probe("DCD_INCREMENTAL_PROJECTION=0", {"DCD_INCREMENTAL_PROJECTION": "0"})
