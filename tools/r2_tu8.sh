#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_u8_obs.py tests/test_gpu_policy.py tests/test_gpu_runner.py tests/test_gpu_graph_rollouts.py tests/test_gpu_sharded_runner.py tests/test_gpu_train_cli.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r2_tu8_tests.log || exit 1
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-leg --no-sweep "$@" 2>gpurun_out/r2_b.err |
      python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']/1e6,1), round(d['ms_per_step'],3))"; }
{ echo "teacher u8"; b; b; b; } 2>&1 | tee gpurun_out/r2_tu8_ab.log
