#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r2_suite.log
