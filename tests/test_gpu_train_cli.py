"""-m gpu: train.py as a user runs it (reference: train.py:39-247) — a child process with the reference's flags:
log files in the reference layout, periodic zero-shot evaluation, checkpointing and resume; and the same driver
under torch.distributed.run with 2 ranks (robust PLR sharded over the ranks; both ranks on the test box's GPU, gloo)."""
import csv
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def _rows(path):
    """logs.csv in the reference layout (filewriter.py:132-229): a '# fields' comment line, then one row per log call;
    the authoritative column list is the last row of fields.csv."""
    names = list(csv.reader(open(os.path.join(os.path.dirname(path), "fields.csv"))))[-1]
    return [dict(zip(names, r)) for r in csv.reader(open(path)) if r and not r[0].startswith("#")]


def test_train_cli_logs_evaluates_checkpoints_and_resumes(tmp_path):
    common = [sys.executable, "train.py", "--xpid=cli", f"--log_dir={tmp_path}", "--env_name=MultiGrid-GoalLastFewerBlocksAdversarial-v0",
              "--ued_algo=paired", "--num_processes=32", "--num_steps=32", "--ppo_epoch=1", "--adv_ppo_epoch=1",
              "--recurrent_adversary_env=True", "--handle_timelimits=True", "--log_interval=1", "--checkpoint=True",
              "--checkpoint_interval=2", "--test_interval=2", "--test_num_episodes=4", "--test_num_processes=4",
              "--test_env_names=MultiGrid-SixteenRooms-v0,MultiGrid-PerfectMazeSmall-v0", "--log_action_complexity=True",
              "--screenshot_interval=3", "--screenshot_batch_size=4"]
    out = _run(common + ["--num_env_steps=4096"])                   # 4 updates of 32 x 32 steps
    d = os.path.join(str(tmp_path), "cli")
    assert os.path.exists(os.path.join(d, "meta.json")) and os.path.exists(os.path.join(d, "model.tar"))
    meta = json.load(open(os.path.join(d, "meta.json")))
    assert meta["args"]["ued_algo"] == "paired" and meta["args"]["num_processes"] == 32
    from PIL import Image                                            # level screenshots at updates 0 and 3 (train.py:204-232)
    assert sorted(os.listdir(os.path.join(d, "screenshots"))) == ["update0.png", "update3.png"]
    shot = Image.open(os.path.join(d, "screenshots", "update3.png"))
    assert shot.size == (4 * (15 * 32 + 2) + 2, 15 * 32 + 4)
    rows = _rows(os.path.join(d, "logs.csv"))
    assert len(rows) == 4 and [int(float(r["steps"])) for r in rows] == [1024, 2048, 3072, 4096]
    for k in ("sps", "mean_agent_return", "agent_value_loss", "adversary_env_value_loss", "mean_env_return", "agent_action_complexity",
              "solved_rate:MultiGrid-SixteenRooms-v0", "test_returns:MultiGrid-PerfectMazeSmall-v0"):
        assert k in rows[0], (k, sorted(rows[0]))
    assert rows[0]["solved_rate:MultiGrid-SixteenRooms-v0"] != ""     # evaluated at update 0
    sd = torch.load(os.path.join(d, "model.tar"), map_location="cpu", weights_only=False)["runner_state_dict"]
    assert set(sd["agent_state_dict"]) == {"agent", "adversary_agent", "adversary_env"} and sd["num_updates"] >= 3
    saved = sd["num_updates"]
    out = _run(common + ["--num_env_steps=7168"])                   # resume: 7 updates in all
    rows = _rows(os.path.join(d, "logs.csv"))
    steps = [int(float(r["steps"])) for r in rows]
    assert steps[-1] == 7168 and steps[4] == (saved + 1) * 1024       # continued from the checkpointed update count


def test_train_cli_checkpoints_by_default_and_after_the_last_update(tmp_path):
    """Reference semantics (train.py:118-125,190-202): saving is ON unless --disable_checkpoint (--checkpoint only means
    'resume'); model.tar is written after the LAST update even off-interval; archives are keyed to
    runner.<checkpoint_basis> (here student_grad_updates, as three shipped configs set it)."""
    common = [sys.executable, "train.py", f"--log_dir={tmp_path}", "--env_name=MultiGrid-MiniGoalLastAdversarial-v0",
              "--ued_algo=domain_randomization", "--use_plr=True", "--use_reset_random_dr=True", "--no_exploratory_grad_updates=True",
              "--level_replay_strategy=positive_value_loss", "--level_replay_score_transform=rank", "--level_replay_prob=0.5",
              "--level_replay_rho=0.1", "--level_replay_seed_buffer_size=100", "--num_processes=32", "--num_steps=32", "--ppo_epoch=1",
              "--handle_timelimits=True", "--log_interval=2", "--test_interval=0", "--num_env_steps=7168",      # 7 updates
              "--checkpoint_interval=100", "--archive_interval=2", "--checkpoint_basis=student_grad_updates"]
    _run(common + ["--xpid=dflt"])
    d = os.path.join(str(tmp_path), "dflt")
    sd = torch.load(os.path.join(d, "model.tar"), map_location="cpu", weights_only=False)["runner_state_dict"]
    assert sd["num_updates"] == 7                                     # the final model, although 7 % 100 != 0
    rows = _rows(os.path.join(d, "logs.csv"))
    assert [int(float(r["steps"])) // 1024 for r in rows] == [1, 3, 5, 7]          # every log_interval and the last update
    grad_updates = sd["student_grad_updates"]
    archives = sorted(int(f[len("model_"):-len(".tar")]) for f in os.listdir(d) if f.startswith("model_") and f.endswith(".tar"))
    assert archives and all(a % 2 == 0 and a <= grad_updates for a in archives)     # numbered by student grad updates
    _run(common + ["--xpid=off", "--disable_checkpoint=True"])
    assert not os.path.exists(os.path.join(str(tmp_path), "off", "model.tar"))


def test_train_cli_two_ranks_robust_plr(tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr=127.0.0.1",
           "--master-port=29641", "train.py", "--xpid=ddp", f"--log_dir={tmp_path}",
           "--env_name=MultiGrid-MiniGoalLastAdversarial-v0", "--ued_algo=domain_randomization", "--use_plr=True",
           "--use_reset_random_dr=True", "--no_exploratory_grad_updates=True", "--level_replay_strategy=positive_value_loss",
           "--level_replay_score_transform=rank", "--level_replay_prob=0.5", "--level_replay_rho=0.2",
           "--level_replay_seed_buffer_size=100", "--staleness_coef=0.3", "--log_plr_buffer_stats=True",
           "--num_processes=32", "--num_steps=64", "--ppo_epoch=1", "--num_env_steps=24576", "--log_interval=1",
           "--test_interval=0", "--handle_timelimits=True"]
    _run(cmd, env={"DCD_DIST_BACKEND": "gloo"})
    rows = _rows(os.path.join(str(tmp_path), "ddp", "logs.csv"))
    assert len(rows) == 6                                            # 24576 / (2 ranks x 32 envs x 64 steps)
    assert int(float(rows[-1]["steps"])) == 6 * 2 * 32 * 64          # job-wide step count (summed over ranks)
    assert all(r["agent_plr_passable_mass"] != "" for r in rows)


def test_wall_limit_checkpoints_leaves_and_a_second_run_resumes(tmp_path):
    """--wall_limit_s (this engine's addition; what tools/dr_pin.sh slices a 250 M-step run with): the loop checkpoints and
    leaves once the budget is spent; the same command line run again continues from that update and finishes the job."""
    common = [sys.executable, "train.py", "--xpid=wall", f"--log_dir={tmp_path}", "--env_name=MultiGrid-MiniGoalLastAdversarial-v0",
              "--ued_algo=domain_randomization", "--use_plr=True", "--level_replay_prob=0.0", "--num_processes=32", "--num_steps=32",
              "--ppo_epoch=1", "--handle_timelimits=True", "--log_interval=1", "--test_interval=0", "--checkpoint=True",
              "--checkpoint_interval=1000", "--num_env_steps=40960"]                       # 40 updates
    out = _run(common + ["--wall_limit_s=0.000001"])                  # spent before the first update: checkpoint of update 0, nothing logged
    assert '"wall_limit_reached_at_update": 0' in out
    d = os.path.join(str(tmp_path), "wall")
    assert os.path.exists(os.path.join(d, "model.tar"))
    out = _run(common + ["--wall_limit_s=3600"])
    assert "wall_limit_reached" not in out
    rows = _rows(os.path.join(d, "logs.csv"))
    assert len(rows) == 40 and int(float(rows[-1]["steps"])) == 40960
    sd = torch.load(os.path.join(d, "model.tar"), map_location="cpu", weights_only=False)["runner_state_dict"]
    assert sd["num_updates"] == 40
