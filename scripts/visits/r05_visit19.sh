#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v19; mkdir -p $OUT
V="auto 128x128/64 128x128/128 128x64/64 128x64/128 64x64/64 64x64/128 256x128/64 256x128/128"
{
echo "## bf16 train bs 32, small-M 1x1 launches (layer3 / layer4): forward with BN statistics"
timeout 600 python scripts/bench_layer.py --variants "$V" \
  bf16:stats:32,256,32,32,1024,1,1,0 bf16:stats:32,1024,32,32,256,1,1,0 bf16:stats:32,512,16,16,2048,1,1,0 bf16:stats:32,2048,16,16,512,1,1,0 \
  bf16:stats:32,512,32,32,1024,1,2,0 bf16:stats:32,1024,16,16,2048,1,2,0
echo "## data gradients into a BatchNorm"
timeout 600 python scripts/bench_layer.py --variants "$V" \
  bf16:bwd+res:32,1024,32,32,256,1,1,0 bf16:bwd:32,256,32,32,1024,1,1,0 bf16:bwd+res:32,2048,16,16,512,1,1,0 bf16:bwd:32,512,16,16,2048,1,1,0 \
  bf16:bwd+res:32,512,64,64,128,1,1,0 bf16:bwd:32,128,64,64,512,1,1,0
} 2>&1 | grep -v "Warn\|amdgpu.ids" | tee $OUT/sweep_small_1x1.txt | cut -c1-150
