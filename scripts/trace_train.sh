#!/bin/bash
# kernel traces of the bf16 bs-32 train leg only: overlapped and serial (ROBOSAT_WGRAD_STREAM=0);  scripts/trace_train.sh TAG
TAG=$1; export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-parity --phase train --dtype bf16 --batch 32 --steps 5 --warmup 2"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -o p -- $B > $OUT/trace_train.log 2>&1; echo "exit $?"
ROBOSAT_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train_serial -o p -- $B > $OUT/trace_train_serial.log 2>&1; echo "exit $?"
cd $REPO
for T in train train_serial; do
  F=$(find $OUT/trace_$T -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/${T}_kernel_stats.csv
done
python scripts/trace_gaps.py $(find $OUT/trace_train -name "*kernel_trace.csv" | head -1) > $OUT/trace_gaps_train.txt 2>&1
python scripts/trace_gaps.py $(find $OUT/trace_train_serial -name "*kernel_trace.csv" | head -1) > $OUT/trace_gaps_train_serial.txt 2>&1
head -3 $OUT/trace_gaps_train.txt
