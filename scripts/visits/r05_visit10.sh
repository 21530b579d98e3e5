#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v10; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_configs.py -m gpu -q --timeout 600 -k "lovasz or cfg5" > $OUT/pytest_lovasz.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_lovasz.log | cut -c1-300
for V in old new; do
  if [ $V = new ]; then unset ROBOSAT_HIP_LIB; else export ROBOSAT_HIP_LIB=$REPO/gpurun_in/librobosat_hip_$V.so; fi
  echo "== $V"; timeout 120 python scripts/bench_lovasz.py 2>&1 | grep -v "Warn\|amdgpu.ids"
done | tee $OUT/lovasz_variants.txt
unset ROBOSAT_HIP_LIB
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $REPO/scripts/bench_lovasz.py > $OUT/prof.log 2>&1; echo "exit $?"
cd $REPO
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/lovasz_kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" -delete
