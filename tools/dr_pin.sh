#!/bin/bash
# End-to-end statistical pin: the reference's shipped DR configuration for MultiGrid
# (train_scripts/grid_configs/minigrid/25_blocks/mg_25b_dr.json -> tests/golden/mg_train_configs.json: 32 envs x 256 steps,
# PLR bookkeeping with replay probability 0, random teacher, 250 M env steps) through train.py, in wall-clock slices that
# checkpoint and resume (a gpurun call lasts at most 20 minutes; runs/ travels to the box, gpurun_out/ comes back).
#   usage: tools/dr_pin.sh <xpid> <wall seconds> [extra train.py flags]
set -e
XPID=${1:-dr_pin}; WALL=${2:-600}; shift 2 || true
mkdir -p runs gpurun_out
python train.py --xpid "$XPID" --log_dir runs --checkpoint true --wall_limit_s "$WALL" \
  --env_name MultiGrid-GoalLastFewerBlocksAdversarial-v0 --ued_algo domain_randomization --use_plr true --level_replay_prob 0.0 \
  --num_processes 32 --num_steps 256 --ppo_epoch 5 --num_mini_batch 1 --lr 0.0001 --gamma 0.995 --entropy_coef 0.01 \
  --handle_timelimits true --recurrent_agent true --recurrent_adversary_env false --recurrent_arch lstm --recurrent_hidden_size 256 \
  --reject_unsolvable_seeds false --log_action_complexity true --log_replay_complexity true --log_plr_buffer_stats true \
  --num_env_steps 250000000 --log_interval 25 --checkpoint_interval 1000 --screenshot_interval 0 \
  --test_interval 250 --test_num_episodes 10 --test_num_processes 2 \
  --test_env_names MultiGrid-SixteenRooms-v0,MultiGrid-Labyrinth-v0,MultiGrid-Maze-v0 "$@" > "gpurun_out/${XPID}_stdout.log" 2>&1
rm -rf "gpurun_out/$XPID"; cp -r "runs/$XPID" "gpurun_out/$XPID"
tail -2 "gpurun_out/${XPID}_stdout.log" | cut -c1-600
