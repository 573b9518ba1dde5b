#!/bin/bash
# same-box A/B: the step kernel of this tree (A) against round 3's mg_kernels.hip (B), whole-pass rate and eager per-launch times
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-leg --no-sweep 2>/dev/null |
      python -c "import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_times']['kernels']['mg_step_kernel'];print('pass M/s', round(d['value']/1e6,1), 'ms', round(d['ms_per_step'],4), 'step avg_launch_us', round(d['roofline']['avg_launch_us'],3), 'median', round(k['median_us'],3), 'accounts_for', round(d['kernel_times']['accounts_for'],3))"; }
echo "A (this tree)"; b; b; b
cp dcd-isaac_amd/csrc/mg_kernels.hip /tmp/mg_kernels_new.hip
cp tools/_ab/mg_kernels_r03.hip dcd-isaac_amd/csrc/mg_kernels.hip
make -j4 -C dcd-isaac_amd/csrc > /dev/null 2>&1
echo "B (round 3's mg_kernels.hip)"; b; b; b
cp /tmp/mg_kernels_new.hip dcd-isaac_amd/csrc/mg_kernels.hip
make -j4 -C dcd-isaac_amd/csrc > /dev/null 2>&1
echo "A again"; b; b
