#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r2_e2e2.log; rm -f $L
N_UPDATES=9 python tools/time_train.py 4096 2>&1 | grep -v amdgpu | grep update >> $L
cat $L
