#!/bin/bash
mkdir -p gpurun_out
python tools/prof_replay_rollout.py 2>&1 | grep -v amdgpu | tee gpurun_out/r2_replay_phases.log
