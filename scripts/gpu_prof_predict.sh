#!/bin/bash
# rocprofv3 kernel stats of the default (fp32 predict) bench.  usage: scripts/gpu_prof_predict.sh TAG [extra bench args]
TAG=${1:-p}; shift
export TMPDIR=/tmp
REPO=$(pwd); mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu-baseline --no-train-leg --steps 10 --warmup 3 --layers-json gpurun_out/layers_predict_$TAG.json "$@" 2>&1 | tail -1 | cut -c1-400
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_predict_$TAG -o p -- python $REPO/bench.py --no-cpu-baseline --no-train-leg --steps 5 --warmup 2 "$@" > $REPO/gpurun_out/rocprof_predict_$TAG.log 2>&1
echo "rocprof exit $?"
find $REPO/gpurun_out/prof_predict_$TAG -name "*kernel_trace*" -size +20M -delete
