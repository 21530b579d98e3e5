#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v9; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_configs.py -m gpu -q --timeout 600 -k "lovasz or cfg5" > $OUT/pytest_lovasz.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_lovasz.log | cut -c1-300
for V in old new sort4096 sort2048; do
  if [ $V = new ]; then unset ROBOSAT_HIP_LIB; else export ROBOSAT_HIP_LIB=$REPO/gpurun_in/librobosat_hip_$V.so; fi
  echo "== $V"; timeout 120 python scripts/bench_lovasz.py 2>&1 | grep -v "Warn\|amdgpu.ids"
done | tee $OUT/lovasz_variants.txt
for V in sort4096 sort2048; do
  ROBOSAT_HIP_LIB=$REPO/gpurun_in/librobosat_hip_$V.so timeout 300 python -m pytest tests/test_gpu_train_ops.py -m gpu -q --timeout 600 -k "lovasz" 2>&1 | tail -1
done
