"""Where a PLR replay rollout's per-step host work goes (robust PLR, B = 4096): wall time per phase of
AdversarialRunner._resample_finished_levels, with a device synchronisation between phases."""
import os, sys, time, json, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import dcd_hip
from arguments import parser
from envs.runners import AdversarialRunner
from util import create_parallel_env, make_agent, make_plr_args, seed
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
flags = json.load(open(os.path.join(ROOT, "tests", "golden", "mg_train_configs.json")))["25_blocks/mg_25b_robust_plr.json"]
args = parser.parse_args([f"--{k}={v}" for k, v in flags.items()])
args.num_processes = B; args.level_replay_rho = 0.2
seed(1)
venv, ued = create_parallel_env(args, device="cuda:0")
agent = make_agent("agent", venv, args, "cuda:0"); teacher = make_agent("adversary_env", venv, args, "cuda:0"); teacher.random()
runner = AdversarialRunner(args=args, venv=venv, agent=agent, ued_venv=ued, adversary_env=teacher, train=True,
                           plr_args=make_plr_args(args, venv.observation_space, venv.action_space), device="cuda:0")
T = collections.defaultdict(float)
def tick(name, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); T[name] += t - t0; return t
def resample(self, st, step, level_sampler, rollout_info):
    torch.cuda.synchronize(); t = time.perf_counter()
    finished = (st.ep_info[step] & dcd_hip.INFO_EPISODE).nonzero().flatten(); idx = finished.cpu().tolist(); t = tick("finished mask -> host", t)
    seeds = level_sampler.sample_replay_levels(len(idx)); t = tick("sample_replay_levels", t)
    T["episodes"] += len(idx)
    if not idx: return
    idx_dev = finished.to(torch.int32); u8 = st.obs_u8 is not None
    out = self.venv.engine.student_buffers(len(idx), u8=u8, f32=not u8); t = tick("alloc obs buffers", t)
    if self.use_byte_encoding:
        rows = self._level_rows(seeds); t = tick("seed -> table rows (H2D of indices)", t)
        enc = self._table.index_select(0, rows); t = tick("gather encodings on device", t)
        self.venv.engine.reset_to_encoding(enc, idx_dev, out); t = tick("reset_to_level kernel", t)
    else:                                                    # levels stored as teacher action strings
        kind, a, b = self.venv._levels_to_device([self.level_store.get_level(s) for s in seeds]); t = tick("level strings -> device", t)
        self.venv.engine.reset_to_actions(a, b, idx_dev, out); t = tick("reset_to_actions kernel", t)
    st.obs_u8["image"][step + 1][finished] = out["image_u8"]; st.obs_u8["direction"][step + 1][finished] = out["dir_u8"]; st._stale[step + 1] = True; t = tick("scatter obs rows", t)
    for i, s in zip(idx, seeds):
        self.current_level_seeds[i] = s; rollout_info["solved_idx"][i] = True
    self._seeds_dev[finished] = torch.tensor(seeds, dtype=torch.int32, device=self.device).view(-1, 1); t = tick("seed bookkeeping", t)
AdversarialRunner._resample_finished_levels = resample
n_rep = 0
for j in range(14):
    T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter(); runner.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if runner.sampled_level_info["level_replay"]:
        n_rep += 1
        if n_rep >= 3:
            print(f"replay update {dt:.3f}s; per-step host phases (256 steps, {int(T['episodes'])} re-sampled episodes):")
            for k, v in T.items():
                if k != "episodes": print(f"   {k:28s} {v*1e3:7.1f} ms")
            break
