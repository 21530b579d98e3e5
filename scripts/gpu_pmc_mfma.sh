#!/bin/bash
# MFMA / LDS / wait counters per kernel for the predict bench and the bf16 train bench (one --pmc pass each, kernel-trace only).
# usage: scripts/gpu_pmc_mfma.sh TAG
TAG=${1:-r01}
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_mfma_$TAG; mkdir -p $OUT; cd /tmp
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
B="python $REPO/bench.py --no-cpu-baseline --no-train-leg --steps 1 --warmup 1"
timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/predict -o p -- $B > $OUT/predict.log 2>&1; echo "exit $?"
ROBOSAT_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/trainbf16 -o p -- $B --phase train --dtype bf16 --batch 32 > $OUT/trainbf16.log 2>&1; echo "exit $?"
find $OUT -name "*kernel_trace.csv" -size +8M -delete
du -sh $OUT
