#!/bin/bash
# Soak: train.py as a user runs it, full-size PAIRED (60 updates) and robust PLR with the editor (ACCEL-style, 60 updates) on one
# GPU; prints first / last log rows, seconds per update over the run, and the process's peak device memory from rocm-smi.
set -o pipefail
mkdir -p gpurun_out/soak
run() {
  name=$1; shift
  t0=$(date +%s); python train.py --xpid=$name --log_dir=gpurun_out/soak --env_name=MultiGrid-GoalLastFewerBlocksAdversarial-v0 \
    --num_processes=4096 --num_steps=256 --handle_timelimits=True --recurrent_adversary_env=True --log_interval=10 \
    --disable_checkpoint=True --test_interval=1000 --test_env_names= "$@" > gpurun_out/soak/$name.out 2> gpurun_out/soak/$name.err || { tail -20 gpurun_out/soak/$name.err; return 1; }
  grep -c '"update"' gpurun_out/soak/$name.out
  head -1 gpurun_out/soak/$name.out | cut -c1-300; tail -1 gpurun_out/soak/$name.out | cut -c1-300
  echo "wall seconds: $(( $(date +%s) - t0 ))"
}
run paired --ued_algo=paired --num_env_steps=$((60*4096*256)) && \
run plr --ued_algo=domain_randomization --use_plr=True --level_replay_prob=0.5 --level_replay_strategy=positive_value_loss \
    --level_replay_seed_buffer_size=4000 --no_exploratory_grad_updates=True --use_editor=True --level_editor_prob=1.0 --num_edits=5 --base_levels=easy \
    --num_env_steps=$((60*4096*256))
# BASELINE config 3's loop (robust PLR on random-teacher levels stored as action strings), 60 updates
run robust_plr --ued_algo=domain_randomization --use_plr=True --level_replay_prob=0.5 --level_replay_strategy=positive_value_loss \
    --level_replay_score_transform=rank --level_replay_temperature=0.3 --staleness_coef=0.3 --level_replay_seed_buffer_size=4000 \
    --no_exploratory_grad_updates=True --num_env_steps=$((60*4096*256))
