#!/bin/bash
# Round-3 soak: train.py as a user runs it, with the speculated replay draws on (default) — robust PLR (BASELINE config 3's loop)
# for 60 full-size updates WITH checkpoints (the sampler is pickled while a prefetch thread may be running) and a resume,
# then PLR with the editor for 30 (the speculation is never used there: the sampler moves again after each edit), PAIRED for 20.
set -o pipefail
mkdir -p gpurun_out/soak3
run() {
  name=$1; shift
  t0=$(date +%s); python train.py --xpid=$name --log_dir=gpurun_out/soak3 --env_name=MultiGrid-GoalLastFewerBlocksAdversarial-v0 \
    --num_processes=4096 --num_steps=256 --handle_timelimits=True --recurrent_adversary_env=True --log_interval=10 \
    --test_interval=1000 --test_env_names= "$@" >> gpurun_out/soak3/$name.out 2>> gpurun_out/soak3/$name.err || { tail -20 gpurun_out/soak3/$name.err; return 1; }
  grep -c '"update"' gpurun_out/soak3/$name.out
  head -1 gpurun_out/soak3/$name.out | cut -c1-260; tail -1 gpurun_out/soak3/$name.out | cut -c1-260
  echo "wall seconds: $(( $(date +%s) - t0 ))"
}
RPLR="--ued_algo=domain_randomization --use_plr=True --level_replay_prob=0.5 --level_replay_strategy=positive_value_loss --level_replay_score_transform=rank --level_replay_temperature=0.3 --staleness_coef=0.3 --level_replay_seed_buffer_size=4000 --level_replay_rho=0.2 --no_exploratory_grad_updates=True --log_plr_buffer_stats=True --checkpoint_interval=10"
run robust_plr $RPLR --num_env_steps=$((40*4096*256)) && \
run robust_plr $RPLR --checkpoint=True --num_env_steps=$((60*4096*256)) && \
run plr_editor --ued_algo=domain_randomization --use_plr=True --level_replay_prob=0.5 --level_replay_strategy=positive_value_loss \
    --level_replay_seed_buffer_size=4000 --no_exploratory_grad_updates=True --use_editor=True --level_editor_prob=1.0 --num_edits=5 --base_levels=easy \
    --disable_checkpoint=True --num_env_steps=$((30*4096*256)) && \
run paired --ued_algo=paired --disable_checkpoint=True --num_env_steps=$((20*4096*256))
rm -f gpurun_out/soak3/*/model*.tar
