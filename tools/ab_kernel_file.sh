#!/bin/bash
# Same-box A/B of the shipped mg_kernels.hip (A) against another version of the file (B), e.g. last round's:
#   git show <commit>:dcd-isaac_amd/csrc/mg_kernels.hip > tools/_ab/old.hip     (here; tools/_ab/ is git-ignored but travels to the GPU box)
#   gpurun -- 'tools/ab_kernel_file.sh tools/_ab/old.hip'
# Whole-pass rate of the replayed graph and the step kernel's eager per-launch times, A, B, A again.  Round 4 used it to find
# (and then to confirm the fix of) a 10 % per-launch regression of the B = 4096 step kernel caused by a semantics-preserving
# refactoring of the file (profiles/r04_ab_step_codegen.log).
ALT=${1:?alternative mg_kernels.hip}
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-leg --no-sweep 2>/dev/null |
      python -c "import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_times']['kernels']['mg_step_kernel'];print('pass M/s', round(d['value']/1e6,1), 'ms', round(d['ms_per_step'],4), 'step avg_launch_us', round(d['roofline']['avg_launch_us'],3), 'median', round(k['median_us'],3), 'accounts_for', round(d['kernel_times']['accounts_for'],3))"; }
echo "A (this tree)"; b; b; b
cp dcd-isaac_amd/csrc/mg_kernels.hip /tmp/mg_kernels_shipped.hip
cp "$ALT" dcd-isaac_amd/csrc/mg_kernels.hip
make -j4 -C dcd-isaac_amd/csrc > /dev/null 2>&1
echo "B ($ALT)"; b; b; b
cp /tmp/mg_kernels_shipped.hip dcd-isaac_amd/csrc/mg_kernels.hip
make -j4 -C dcd-isaac_amd/csrc > /dev/null 2>&1
echo "A again"; b; b
