#!/bin/bash
mkdir -p gpurun_out
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-leg --no-sweep "$@" 2>gpurun_out/r2_b.err |
      python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']/1e6,1), round(d['ms_per_step'],3), d['config']['episode_records_last_pass'])"; }
{ echo "fork scans"; b; b; b; echo "serial scans"; DCD_BENCH_FORK_SCANS=0 b; DCD_BENCH_FORK_SCANS=0 b; } 2>&1 | tee gpurun_out/r2_fork_ab.log
tail -3 gpurun_out/r2_b.err
