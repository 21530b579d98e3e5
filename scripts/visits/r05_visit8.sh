#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v8; mkdir -p $OUT
timeout 120 python scripts/bench_lovasz.py 2>&1 | grep -v Warn | tee $OUT/lovasz_8192.txt
ROBOSAT_HIP_LIB=$REPO/gpurun_in/librobosat_hip_sort4096.so timeout 120 python scripts/bench_lovasz.py 2>&1 | grep -v Warn | tee $OUT/lovasz_4096.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $REPO/scripts/bench_lovasz.py > $OUT/prof.log 2>&1; echo "exit $?"
cd $REPO
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/lovasz_kernel_stats.csv && head -14 $OUT/lovasz_kernel_stats.csv | cut -c1-160
find $OUT/prof -name "*kernel_trace.csv" -delete
