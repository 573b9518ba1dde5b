#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_runner.py -x -q -m gpu -k "concurrent or same_seed" 2>&1 | tail -25 | tee gpurun_out/r2_tests9.log
