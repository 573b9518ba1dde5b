#!/bin/bash
# Same-box A/B of a kernel change (box-to-box noise is +-3 %, more than most micro-optimisations are worth):
#   gpurun -- 'bash tools/ab_bench.sh "<sed expression turning variant A into variant B>" <file>'
# Runs bench.py 3x with the tree as shipped (A), applies the sed to <file>, rebuilds the library on the box and runs
# bench.py 3x again (B).  Prints whole-pass env-steps/s (millions) and the step kernel's average launch (us).
set -e
b() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-train-leg --no-sweep 2>/dev/null |
      python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']/1e6,1), round(d['roofline']['avg_launch_us'],3))"; }
echo "A (as shipped)"; b; b; b
sed -i "$1" "$2"
make -j4 -C dcd-isaac_amd/csrc > /dev/null
echo "B ($1)"; b; b; b
