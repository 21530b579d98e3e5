#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v17; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity"
cd /tmp
ROBOSAT_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_f32 -o p -- $B --phase train --dtype fp32 --batch 8 --steps 5 --warmup 2 > $OUT/trace_f32.log 2>&1; echo "exit $?"
cd $REPO
F=$(find $OUT/trace_f32 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/train_fp32_bs8_serial_kernel_stats.csv
find $OUT/trace_f32 -name "*kernel_trace.csv" -delete
