#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_u8_obs.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r2_obs8_tests.log || exit 1
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-leg "$@" 2>gpurun_out/r2_b.err |
      python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']/1e6,1), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_us'],3), d['roofline']['kernel'], [(x['envs'], round(x['us_per_launch'],1), round(x['frac'],3)) for x in d.get('roofline_large_b',[])])"; }
{ echo "u8 no-float-lut"; b; b; b; } 2>&1 | tee gpurun_out/r2_obs8_ab.log
