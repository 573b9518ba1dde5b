#!/usr/bin/env python3
"""Zero-shot evaluation on the fixed MultiGrid mazes, with the reference's Evaluator interface
(reference: eval.py:148-329): `Evaluator(env_names, num_processes, num_episodes, device)`,
`.evaluate(agent, deterministic)` -> {'solved_rate:<env>': ..., 'test_returns:<env>': ...}, `.close()`.

Each env name gets its own GPU batch (HipAdversarialVecEnv over the maze description of
envs/multigrid/maze.py).  The episode loop keeps the reference's semantics — LSTM state zeroed for finished
episodes (eval.py:303-305), the first `num_episodes` finished episodes counted in env order per step — but runs
on device tensors: one small D2H copy of the per-step episode flags instead of B pipe round trips.
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import dcd_hip  # noqa: E402
from envs.registration import make as gym_make  # noqa: E402
from envs.wrappers import HipAdversarialVecEnv  # noqa: E402
from util import is_discrete_actions  # noqa: E402


class Evaluator:
    def __init__(self, env_names, num_processes, num_episodes=10, record_video=False, device="cuda:0", **kwargs):
        if record_video:
            raise NotImplementedError("video recording is UI, outside the GPU hot path")
        self.kwargs = kwargs
        self.env_names = list(env_names)
        self.num_processes = num_processes
        self.num_episodes = num_episodes
        self.device = torch.device(device)
        self.solved_threshold = 0
        self._fused = {}
        self.venv = {}
        for name in self.env_names:
            if not (name.startswith("MultiGrid") or name.startswith("MiniGrid")):           # eval.py:208
                raise ValueError(f"{name}: only the MultiGrid evaluation families are built (SURVEY §8f-2)")
            make_fn = [lambda n=name: gym_make(n)] * num_processes
            self.venv[name] = HipAdversarialVecEnv(make_fn, adversary=False, is_eval=True, device=device)
        self.is_discrete_actions = is_discrete_actions(self.venv[self.env_names[0]])

    def _fused_actor(self, net, B):
        from dcd_hip.policy import StudentActor
        key = (id(net), B)
        if key not in self._fused:
            self._fused[key] = StudentActor.try_create(net, B, seed=0xE7A1)
        fa = self._fused[key]
        return None if (fa is None or net.random) else fa

    def get_stats_keys(self):
        keys = []
        for name in self.env_names:
            keys += [f"solved_rate:{name}", f"test_returns:{name}"]
        return keys

    def close(self):
        for v in self.venv.values():
            v.close()

    def evaluate(self, agent, deterministic=False, show_progress=False, render=False, accumulator="mean"):
        B, dev = self.num_processes, self.device
        env_returns, env_solved = {}, {}
        net = agent.algo.actor_critic
        for name, venv in self.venv.items():
            returns, solved = [], 0
            obs = venv.reset()
            H = net.recurrent_hidden_state_size
            hxs = torch.zeros(B, H, device=dev)
            if net.is_recurrent and net.rnn.arch == "lstm":
                hxs = (hxs, torch.zeros_like(hxs))
            masks = torch.ones(B, 1, device=dev)
            out = venv.engine.step_buffers(u8=False)
            fused = self._fused_actor(net, B)
            if fused is not None:                                    # policy step on the HIP path (dcd_hip.policy)
                fused.begin_rollout(1 << 20)
                hc = [torch.zeros(B, 2 * H, device=dev) for _ in range(2)]
                fo = dict(value=torch.zeros(B, 1, device=dev), action=torch.zeros(B, 1, dtype=torch.int64, device=dev),
                          logp=torch.zeros(B, 1, device=dev), log_dist=torch.zeros(B, net.action_space.n, device=dev))
                t = 0
            while len(returns) < self.num_episodes:
                if fused is not None:
                    fused.counter = t
                    fused.act_into(obs["image"], obs["direction"], hc[t & 1], masks, hc[(t + 1) & 1], fo["value"], fo["action"],
                                   fo["logp"], fo["log_dist"], deterministic=deterministic)
                    action, t = fo["action"], t + 1
                else:
                    with torch.no_grad():
                        _, action, _, hxs = agent.act(obs, hxs, masks, deterministic=deterministic)
                venv.step_env_device(action.view(-1), out)
                obs = {"image": out["obs"]["image_f32"], "direction": out["obs"]["dir_f32"]}
                masks = out["masks"].view(B, 1).clone()
                finished = (out["info"] & dcd_hip.INFO_EPISODE) != 0
                idx = finished.nonzero().flatten()
                if idx.numel():
                    if agent.is_recurrent:                       # eval.py:303-305
                        keep = (~finished).float().view(B, 1)
                        if fused is not None:
                            hc[t & 1].mul_(keep)
                        else:
                            hxs = tuple(h * keep for h in hxs) if isinstance(hxs, tuple) else hxs * keep
                    if venv.is_procedural:                       # the worker's env.reset() draws a new level
                        first = venv.new_levels(idx.cpu().tolist())
                        obs["image"][idx] = first["image"]
                        obs["direction"][idx] = first["direction"]
                    for r in out["ep_ret"][idx].cpu().tolist():  # env order, as the reference's loop over infos
                        returns.append(r)
                        solved += int(r > self.solved_threshold)
                        if len(returns) >= self.num_episodes:
                            break
            venv.raise_errors()
            env_returns[name], env_solved[name] = returns, solved
        stats = {}
        for name in self.env_names:
            if accumulator == "mean":
                stats[f"solved_rate:{name}"] = env_solved[name] / self.num_episodes
                stats[f"test_returns:{name}"] = float(np.mean(env_returns[name]))
            else:
                stats[f"test_returns:{name}"] = env_returns[name]
        return stats


if __name__ == "__main__":
    import argparse
    import json
    from arguments import parser as train_parser
    from util import create_parallel_env, make_agent

    ap = argparse.ArgumentParser()
    ap.add_argument("--model_tar", type=str, default=None, help="checkpoint written by train.py (runner_state_dict)")
    ap.add_argument("--env_names", type=str, default="MultiGrid-SixteenRooms-v0,MultiGrid-Labyrinth-v0,MultiGrid-Maze-v0")
    ap.add_argument("--num_processes", type=int, default=256)
    ap.add_argument("--num_episodes", type=int, default=100)
    ap.add_argument("--deterministic", action="store_true")
    a = ap.parse_args()
    targs = train_parser.parse_args([])
    targs.num_processes = a.num_processes
    venv, _ = create_parallel_env(targs, device="cuda:0")
    agent = make_agent("agent", venv, targs, "cuda:0")
    if a.model_tar:
        from util import load_checkpoint_safe                      # allow-list unpickler: nothing the file names is executed
        sd = load_checkpoint_safe(a.model_tar, map_location="cuda:0")
        agent.algo.actor_critic.load_state_dict(sd["runner_state_dict"]["agent_state_dict"]["agent"])
    ev = Evaluator(a.env_names.split(","), a.num_processes, a.num_episodes, device="cuda:0")
    print(json.dumps(ev.evaluate(agent, deterministic=a.deterministic)))
    ev.close()
