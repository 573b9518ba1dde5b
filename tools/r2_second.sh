#!/bin/bash
set -e
mkdir -p gpurun_out
rm -f gpurun_out/graph_assert_probe.log
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-leg --no-sweep "$@" 2>gpurun_out/r2_b.err |
      python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']/1e6,1), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_us'],3), d['roofline']['kernel'])"; }
{
echo "pair f32"; b; b
echo "pair u8"; b --obs u8; b --obs u8
} 2>&1 | tee gpurun_out/r2_ab2.log
python tools/graph_assert_probe.py all 4096 256 2>&1 | tail -40
