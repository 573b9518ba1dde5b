#!/bin/bash
# robust PLR at 4096 envs: seconds per update with the next update's first draws prefetched under the PPO kernels (default) and
# drawn in program order (DCD_PREFETCH_DRAWS=0), on one GPU as is and with rank 0's host load of an 8-rank job (SIM_WORLD=8:
# 8 x as many draws per request)
mkdir -p gpurun_out
for sim in 1 8; do for pf in 1 0; do
  echo "=== SIM_WORLD=$sim DCD_PREFETCH_DRAWS=$pf"
  SIM_WORLD=$sim DCD_PREFETCH_DRAWS=$pf NO_PROFILE=1 timeout -k 10 200 python tools/time_plr.py 2>&1 | grep '"update"' | tail -8
done; done
