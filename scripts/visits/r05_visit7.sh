#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v7; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_configs.py -m gpu -q --timeout 600 -k "lovasz or cfg5" > $OUT/pytest_lovasz.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_lovasz.log | cut -c1-300
timeout 120 python scripts/bench_lovasz.py 2>&1 | grep -v Warn | tee $OUT/lovasz_new.txt
ROBOSAT_HIP_LIB=$REPO/gpurun_in/librobosat_hip_old.so timeout 120 python scripts/bench_lovasz.py 2>&1 | grep -v Warn | tee $OUT/lovasz_old.txt
