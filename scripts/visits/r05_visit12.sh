#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v12; mkdir -p $OUT
for X in 1 2 1 2; do
  echo "== RS_LOVASZ_XCD=$X"; RS_LOVASZ_XCD=$X timeout 120 python scripts/bench_lovasz.py 2>&1 | grep -v "Warn\|amdgpu.ids"
done | tee $OUT/lovasz_xcd.txt
RS_LOVASZ_XCD=2 timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q --timeout 600 -k "lovasz" 2>&1 | tail -1
