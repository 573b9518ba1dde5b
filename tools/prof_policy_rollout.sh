#!/bin/bash
# Evidence trail for S_roll: rocprofv3 kernel stats of the PAIRED students' lock-step rollout with the policy in
# the loop (tools/time_pair_rollout.py: 4096 envs x 256 steps x 2 students, one HIP graph), reduced to per-kernel average
# duration, algorithmic bytes / flops per launch and achieved GB/s / TFLOP/s.
#   usage: tools/prof_policy_rollout.sh <tag>   -> gpurun_out/<tag>_policy_rollout_kernel_stats.csv, gpurun_out/<tag>_policy_rollout.json
set -o pipefail
TAG=${1:-r04}
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/prof/roll
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/roll -- python3 tools/time_pair_rollout.py > gpurun_out/prof_roll.out 2>&1 || { tail -5 gpurun_out/prof_roll.out; exit 1; }
f=$(find gpurun_out/prof/roll -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${TAG}_policy_rollout_kernel_stats.csv
python3 tools/policy_rollout_roofline.py gpurun_out/${TAG}_policy_rollout_kernel_stats.csv "$(grep both_rollouts_ms gpurun_out/prof_roll.out | tail -1)" > gpurun_out/${TAG}_policy_rollout.json
cat gpurun_out/${TAG}_policy_rollout.json
