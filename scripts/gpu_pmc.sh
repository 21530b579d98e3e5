#!/bin/bash
# PMC counter passes over the predict bench (each pass in its own rocprofv3 run, no trace domains besides kernel-trace)
TAG=${1:-pmc}
export TMPDIR=/tmp
REPO=$(pwd); mkdir -p gpurun_out; cd /tmp
run() { # name counters...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmc_$TAG/$name -o p -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg > $REPO/gpurun_out/pmc_${TAG}_$name.log 2>&1
  echo "pass $name exit $?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_WAVES
run tcc1 FETCH_SIZE TCC_HIT_sum
run tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
cd $REPO
python scripts/pmc_summary.py gpurun_out/pmc_$TAG | tee gpurun_out/pmc_${TAG}_summary.txt
