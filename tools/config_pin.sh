#!/bin/bash
# Any of the reference's shipped MultiGrid configurations (train_scripts/grid_configs/minigrid/*/*.json -> tests/golden/mg_train_configs.json)
# through train.py in wall-clock slices, as tools/dr_pin.sh / tools/robust_plr_pin.sh.
#   usage: tools/config_pin.sh <config key> <xpid> <wall seconds>
set -e
CFG=${1:?config key of tests/golden/mg_train_configs.json, e.g. 25_blocks/mg_25b_paired.json}; XPID=${2:?xpid}; WALL=${3:-600}; shift 3 || true
mkdir -p runs gpurun_out
python - "$CFG" "$XPID" "$WALL" "$@" > "gpurun_out/${XPID}_stdout.log" 2>&1 <<'PY'
import json, os, subprocess, sys
cfg_key, xpid, wall, extra = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
cfg = json.load(open("tests/golden/mg_train_configs.json"))[cfg_key]
skip = {"log_dir", "archive_interval", "checkpoint"}
flags = []
for k, v in cfg.items():
    if k in skip:
        continue
    flags += [f"--{k}", str(v).lower() if isinstance(v, bool) else str(v)]
cmd = [sys.executable, "train.py", "--xpid", xpid, "--log_dir", "runs", "--checkpoint", "true", "--wall_limit_s", wall,
       "--checkpoint_interval", "1000", "--screenshot_interval", "0", "--test_interval", "250", "--test_num_episodes", "10", "--test_num_processes", "2",
       "--test_env_names", "MultiGrid-SixteenRooms-v0,MultiGrid-Labyrinth-v0,MultiGrid-Maze-v0"] + flags + extra
print(" ".join(cmd), flush=True)
sys.exit(subprocess.call(cmd))
PY
rm -rf "gpurun_out/$XPID"; cp -r "runs/$XPID" "gpurun_out/$XPID"
tail -2 "gpurun_out/${XPID}_stdout.log" | cut -c1-400
