#!/usr/bin/env python3
"""train.py — DCD training driver with the reference's flags and loop shape (reference: train.py:38-238),
on the MI355X engine.  One process per GPU (launch with torch.distributed.run for several).

  python train.py --env_name MultiGrid-GoalLastFewerBlocksAdversarial-v0 --ued_algo paired \
      --num_processes 4096 --num_steps 256 --handle_timelimits true --recurrent_adversary_env true \
      --num_env_steps 100000000 --log_interval 1

Logs go to <log_dir>/<xpid>/{logs.csv, fields.csv, meta.json} in the reference's layout (util/filewriter.py) and
to stdout as JSON lines; checkpoints use the reference's `runner_state_dict` layout through util.safe_checkpoint;
every --test_interval updates the student is evaluated zero-shot on --test_env_names (eval.Evaluator, the
15x15 mazes); every --screenshot_interval updates the first --screenshot_batch_size levels are written to
<log_dir>/<xpid>/screenshots/ (train.py:204-232; host-side renderer, envs/multigrid/rendering.py).
"""
import json
import os
import sys
import timeit

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dcd-isaac_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from arguments import parser  # noqa: E402
from envs.runners import AdversarialRunner  # noqa: E402
from util import create_parallel_env, load_checkpoint_safe, make_agent, make_plr_args, safe_checkpoint, save_images, seed  # noqa: E402
from util import distributed as D  # noqa: E402
from util.filewriter import FileWriter  # noqa: E402


def main():
    os.environ["OMP_NUM_THREADS"] = "1"
    args = parser.parse_args()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(args.device or f"cuda:{local_rank % max(torch.cuda.device_count(), 1)}")
    torch.cuda.set_device(device)
    world = D.init(device=device)
    rank = D.rank()
    seed(args.seed + rank)
    pg = torch.distributed.group.WORLD if world > 1 else None
    venv, ued_venv = create_parallel_env(args, device=device, env_index_base=rank * args.num_processes)
    agent = make_agent("agent", venv, args, device, process_group=pg)
    adversary_agent = adversary_env = None
    is_training_env = args.ued_algo in ["paired", "flexible_paired", "minimax"]
    is_paired = args.ued_algo in ["paired", "flexible_paired"]
    if is_paired:
        adversary_agent = make_agent("adversary_agent", venv, args, device, process_group=pg)
    if is_training_env or args.ued_algo == "domain_randomization":
        adversary_env = make_agent("adversary_env", venv, args, device, process_group=pg)
    if args.ued_algo == "domain_randomization" and args.use_plr and not args.use_reset_random_dr:
        adversary_env.random()
    plr_args = make_plr_args(args, venv.observation_space, venv.action_space) if args.use_plr else None
    runner = AdversarialRunner(args=args, venv=venv, agent=agent, ued_venv=ued_venv, adversary_agent=adversary_agent,
                               adversary_env=adversary_env, flexible_protagonist=False, train=True, plr_args=plr_args,
                               device=device, process_group=pg)
    num_updates = int(args.num_env_steps) // args.num_steps // (args.num_processes * world)
    ckpt = os.path.expandvars(os.path.expanduser(os.path.join(args.log_dir, args.xpid, "model.tar")))
    writer = FileWriter(xpid=args.xpid, xp_args=vars(args), rootdir=args.log_dir) if rank == 0 else None
    evaluator = None
    if rank == 0 and args.test_env_names and args.test_interval > 0:
        from eval import Evaluator
        names = [n for n in args.test_env_names.split(",") if n]
        try:
            evaluator = Evaluator(names, num_processes=args.test_num_processes, num_episodes=args.test_num_episodes, device=device)
        except NotImplementedError as e:                  # a family the engine does not serve (e.g. the 3-agent cluttered grid)
            print(f"evaluation disabled: {e}", flush=True)

    def checkpoint(index=None):                            # train.py:118-125: always on unless --disable_checkpoint
        if args.disable_checkpoint or rank != 0:
            return
        os.makedirs(os.path.dirname(ckpt), exist_ok=True)
        safe_checkpoint({"runner_state_dict": runner.state_dict()}, ckpt, index=index, archive_interval=args.archive_interval)

    last_logged_at_restart = -1
    if args.checkpoint and os.path.exists(ckpt):           # --checkpoint only means "resume a preempted job" (train.py:129-134)
        runner.load_state_dict(load_checkpoint_safe(ckpt, map_location=device)["runner_state_dict"])
        if writer is not None:
            last_logged_at_restart = writer.latest_tick()  # ticks are 0-indexed updates: rows already in logs.csv are not repeated
        # only rank 0 reads logs.csv: every rank must skip the same updates (the evaluation below is a collective decision)
        last_logged_at_restart = int(D.broadcast_ints([last_logged_at_restart], 1, group=pg, device=device)[0]) if world > 1 else last_logged_at_restart
    elif args.xpid_finetune and not os.path.exists(ckpt):  # first iteration of a fine-tuning run (train.py:135-141)
        base = os.path.expandvars(os.path.expanduser(os.path.join(args.log_dir, args.xpid_finetune, f"{args.model_finetune}.tar")))
        sd = load_checkpoint_safe(base, map_location=device)["runner_state_dict"]    # a foreign file: nothing it names is executed
        runner.agents["agent"].algo.actor_critic.load_state_dict(sd["agent_state_dict"]["agent"])
        runner.agents["agent"].algo.optimizer.load_state_dict(sd["optimizer_state_dict"]["agent"])
    idx_at_start = getattr(runner, args.checkpoint_basis)
    t0 = t_start = timeit.default_timer()
    for j in range(runner.num_updates, num_updates):
        if args.wall_limit_s > 0:
            # (this engine's addition) a job with a wall-clock budget: checkpoint and leave; `--checkpoint true` resumes it.
            # Rank 0's clock decides for every rank (one small broadcast per update, only when the flag is set)
            over = timeit.default_timer() - t_start > args.wall_limit_s
            if world > 1:
                over = bool(D.broadcast_ints([int(over)], 1, group=pg, device=device)[0])
            if over:
                checkpoint(getattr(runner, args.checkpoint_basis))
                if rank == 0:
                    print(json.dumps({"wall_limit_reached_at_update": j}), flush=True)
                break
        stats = D.reduce_stats(runner.run())
        if runner.num_updates <= last_logged_at_restart:   # train.py:165-166: neither logged, checkpointed nor screenshot again
            continue
        if j % args.log_interval == 0 or j == num_updates - 1:
            test_stats = {}
            if evaluator is not None and (j % args.test_interval == 0 or j == num_updates - 1):
                test_stats = evaluator.evaluate(runner.agents["agent"])
            if rank == 0:
                torch.cuda.synchronize()
                now = timeit.default_timer()
                n = 1 if j == 0 else args.log_interval                           # train.py:182-184
                sps = n * args.num_processes * world * args.num_steps / (now - t0)
                t0 = now
                row = {k: (float(v) if isinstance(v, (int, float, np.floating, np.integer)) else None)
                       for k, v in stats.items() if not hasattr(v, "__len__")}
                row.update(sps=sps)
                row.update(test_stats)                                           # sps column before the test stats
                writer.log(dict(row))
                print(json.dumps(dict(row, update=j)), flush=True)
        # train.py:190-202: keyed to runner.<checkpoint_basis> (num_updates or student_grad_updates); model.tar at every
        # checkpoint_interval and after the LAST update, numbered copies at every archive_interval
        idx = getattr(runner, args.checkpoint_basis)
        if idx != idx_at_start:                            # (the reference compares with the value before the loop, :158,190-192)
            if j == num_updates - 1 or (runner.num_updates > 0 and idx % args.checkpoint_interval == 0):
                checkpoint(idx)
            elif runner.num_updates > 0 and args.archive_interval > 0 and idx % args.archive_interval == 0:
                checkpoint(idx)
        if rank == 0 and args.screenshot_interval > 0 and j % args.screenshot_interval == 0:     # train.py:169-171,204-232
            shot_dir = os.path.join(os.path.dirname(ckpt), "screenshots")
            os.makedirs(shot_dir, exist_ok=True)
            info = runner.sampled_level_info
            name = (f"update{j}-replay{info['level_replay']}-n_edits{info['num_edits'][0]}.png"
                    if args.use_editor and info else f"update{j}.png")
            venv.reset_agent()
            save_images(venv.get_images(range(min(args.screenshot_batch_size, args.num_processes))),
                        os.path.join(shot_dir, name), normalize=True, channels_first=False)
    if writer is not None:
        writer.close()
    if evaluator is not None:
        evaluator.close()
    venv.close()


if __name__ == "__main__":
    main()
