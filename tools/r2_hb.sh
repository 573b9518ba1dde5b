#!/bin/bash
mkdir -p gpurun_out
python tools/ppo_hostbound.py 2>&1 | grep -v amdgpu > gpurun_out/ppo_hostbound.log; grep -E "update wall" gpurun_out/ppo_hostbound.log; head -40 gpurun_out/ppo_hostbound.log | cut -c1-200
