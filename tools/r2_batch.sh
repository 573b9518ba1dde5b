#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_runner.py tests/test_gpu_u8_obs.py tests/test_gpu_graph_rollouts.py -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r2_batch_tests.log
