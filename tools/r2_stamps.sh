#!/bin/bash
mkdir -p gpurun_out
python tools/stamp_step.py 4096 f32 2>&1 | grep -v amdgpu | head -30 > gpurun_out/stamps_f32.log
python tools/stamp_step.py 4096 u8 2>&1 | grep -v amdgpu | head -30 > gpurun_out/stamps_u8.log
head -12 gpurun_out/stamps_f32.log; echo; head -12 gpurun_out/stamps_u8.log
