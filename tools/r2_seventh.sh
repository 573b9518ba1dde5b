#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_runner.py tests/test_gpu_graph_rollouts.py tests/test_gpu_policy.py -x -q -m gpu 2>&1 | tail -40 | tee gpurun_out/r2_tests7.log
