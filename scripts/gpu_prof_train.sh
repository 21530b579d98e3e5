#!/bin/bash
# rocprofv3 kernel-trace stats of the train bench.  usage: scripts/gpu_prof_train.sh TAG DTYPE BATCH
TAG=${1:-t}; DT=${2:-bf16}; BS=${3:-32}
export TMPDIR=/tmp
REPO=$(pwd); mkdir -p gpurun_out
echo "== bench train $DT bs$BS"
timeout 900 python bench.py --phase train --dtype $DT --batch $BS --steps 5 --warmup 2 --no-cpu-baseline --layers-json gpurun_out/layers_train_${DT}_bs${BS}_$TAG.json > gpurun_out/bench_train_${DT}_bs${BS}_$TAG.log 2>&1
echo "exit $?"; tail -1 gpurun_out/bench_train_${DT}_bs${BS}_$TAG.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_train_${DT}_$TAG -o train -- python $REPO/bench.py --phase train --dtype $DT --batch $BS --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/rocprof_train_${DT}_$TAG.log 2>&1
echo "rocprof exit $?"
cd $REPO
find gpurun_out/prof_train_${DT}_$TAG -name "*kernel_stats*" | head -1 | xargs -I{} sh -c 'head -40 {}'
find gpurun_out/prof_train_${DT}_$TAG -name "*kernel_trace*" -size +20M -delete
