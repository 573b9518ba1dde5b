#!/bin/bash
# Where do the wave cycles of the wide step kernel go?  SQ counter passes (8 SQ slots per pass, never combined with tracing) over
# `tools/sweep_only.py <B>` (dcd_mg_step alone, HBM-bound regime), reduced per kernel by tools/pmc_reduce.py.
#   usage: tools/pmc_step_bottleneck.sh <tag> [B ...]      -> gpurun_out/<tag>_pmc_sq.json
set -o pipefail
TAG=${1:-r04}; shift || true
SIZES=${@:-1048576}
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
P3="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
P4="FETCH_SIZE"          # TCC passes on their own (MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2)
P5="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rm -rf gpurun_out/prof/sq$i
  rocprofv3 --pmc $P --output-format csv -d gpurun_out/prof/sq$i -- python3 tools/sweep_only.py $SIZES > gpurun_out/prof_sq$i.out 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/prof_sq$i.out; }
done
python3 tools/pmc_reduce.py "mg_step_" gpurun_out/prof/sq1 gpurun_out/prof/sq2 gpurun_out/prof/sq3 gpurun_out/prof/sq4 gpurun_out/prof/sq5 > gpurun_out/${TAG}_pmc_sq.json
cat gpurun_out/${TAG}_pmc_sq.json
