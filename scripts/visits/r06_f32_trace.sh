#!/bin/bash
# fp32 bs-8 train step: kernel stats (overlapped and serial) and the per-queue picture (scripts/trace_gaps.py)
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/f32trace; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --phase train --dtype fp32 --batch 8 --no-parity"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o p -- $B --steps 4 --warmup 2 > $OUT/trace.log 2>&1; echo "exit $?"
ROBOSAT_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ts -o p -- $B --steps 4 --warmup 2 > $OUT/trace_serial.log 2>&1; echo "exit $?"
cd $REPO
F=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
[ -n "$F" ] && python scripts/trace_gaps.py $F conv_igemm_f32 > $OUT/train_f32_bs8_trace_gaps.txt 2>&1
cp $(find $OUT/t -name "*kernel_stats.csv" | head -1) $OUT/train_f32_bs8_kernel_stats.csv
cp $(find $OUT/ts -name "*kernel_stats.csv" | head -1) $OUT/train_f32_bs8_serial_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -size +8M -delete
head -30 $OUT/train_f32_bs8_trace_gaps.txt | cut -c1-200
