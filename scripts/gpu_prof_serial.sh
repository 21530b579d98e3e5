#!/bin/bash
# rocprofv3 kernel stats of the bf16 train bench with the wgrad side stream off (clean per-kernel durations).
TAG=${1:-s}; BS=${2:-32}
export TMPDIR=/tmp ROBOSAT_WGRAD_STREAM=0
REPO=$(pwd); mkdir -p gpurun_out
timeout 600 python bench.py --phase train --dtype bf16 --batch $BS --steps 5 --warmup 2 --no-cpu-baseline --layers-json gpurun_out/layers_train_serial_$TAG.json 2>&1 | tail -1 | cut -c1-300
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_serial_$TAG -o train -- python $REPO/bench.py --phase train --dtype bf16 --batch $BS --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/rocprof_serial_$TAG.log 2>&1
echo "rocprof exit $?"
find $REPO/gpurun_out/prof_serial_$TAG -name "*kernel_trace*" -size +20M -delete
