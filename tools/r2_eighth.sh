#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python tools/ppo_determinism.py 256 64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ppo_determinism.log
python -m pytest tests/test_gpu_train_cli.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r2_tests8.log
