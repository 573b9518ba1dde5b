#!/bin/bash
# round-2 first GPU call: new parity tests, then same-box A/B of the pair launch
set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_scans.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r2_tests1.log
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-leg --no-sweep "$@" 2>gpurun_out/r2_b.err |
      python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']/1e6,1), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_us'],3), d['roofline']['kernel'])"; }
{
echo "no-pair"; b --no-pair; b --no-pair
echo "pair epb16"; b; b; b
echo "pair epb32"; DCD_STEP_EPB=32 b; DCD_STEP_EPB=32 b
} 2>&1 | tee gpurun_out/r2_ab1.log
