#!/bin/bash
# Round-5 box visit 32: do the short-K / wide-output 1x1 launches care whether a block writes WHOLE output rows (BN = Cout) or half rows?
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v32; mkdir -p $OUT
V="auto 128x128/64 128x128/128 256x128/64 256x256/128 256x256/64 128x64/64"
{
timeout 600 python scripts/bench_layer.py --iters 30 --variants "$V" \
  bf16:stats:32,64,128,128,256,1,1,0 bf16:stats:32,128,64,64,512,1,1,0 bf16:stats:32,256,128,128,64,1,1,0 bf16:stats:32,256,128,128,128,1,1,0 \
  bf16:conv+res:32,64,128,128,256,1,1,0
} 2>&1 | grep -v "Warn\|amdgpu.ids" | tee $OUT/wide_rows.txt
echo "=== done ($(date +%T))"
