#!/bin/bash
# rehearsal of the multi-rank bench path with 2 ranks on the one GPU (gloo: RCCL refuses two ranks per device)
mkdir -p gpurun_out
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo > gpurun_out/bench_2rank_r2.log 2>&1; echo "rc $?"; tail -3 gpurun_out/bench_2rank_r2.log | cut -c1-600
