#!/bin/bash
# cost of TENSILE_STREAMK_DATA_PARALLEL=1 for the whole process at the benchmark shape (4096 x 256): concurrent and sequential updates
mkdir -p gpurun_out
echo "=== time_train, concurrent, data-parallel GEMMs process-wide"; DCD_CONCURRENT_PPO=1 TENSILE_STREAMK_DATA_PARALLEL=1 N_UPDATES=6 timeout -k 10 200 python tools/time_train.py 2>&1 | grep "run_s\|S_roll\|rollout" | cut -c1-150
echo "=== time_train, sequential, data-parallel GEMMs process-wide"; TENSILE_STREAMK_DATA_PARALLEL=1 DCD_CONCURRENT_PPO=0 N_UPDATES=5 timeout -k 10 200 python tools/time_train.py 2>&1 | grep "run_s\|S_roll\|rollout" | cut -c1-150
echo "=== time_train, sequential, default GEMMs"; DCD_CONCURRENT_PPO=0 N_UPDATES=5 timeout -k 10 200 python tools/time_train.py 2>&1 | grep "run_s\|S_roll\|rollout" | cut -c1-150
echo done
