#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-leg "$@" 2>gpurun_out/r2_b.err |
      python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']/1e6,1), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_us'],3), [(x['envs'], round(x['us_per_launch'],1), round(x['frac'],3)) for x in d.get('roofline_large_b',[])])"; }
{ echo "B: nt stores"; b; b; b
  sed -i 's/#define DCD_NT_STORES 1/#define DCD_NT_STORES 0/' dcd-isaac_amd/csrc/mg_kernels.hip; make -j4 -C dcd-isaac_amd/csrc > /dev/null
  echo "A: plain stores"; b; b; b; } 2>&1 | tee gpurun_out/r2_nt_ab.log
