#!/bin/bash
# Round-5 box visit 28: tile x row sweep over the small-M fp32 launches of the bs-8 fp32 training step (layer3 / layer4: 32 launches of
# conv_igemm_f32<64x64,r128> at ~80 us = 2.5 of 28.9 ms serial, 34 % of the fp32 MFMA peak).
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v28; mkdir -p $OUT
V="auto 64x64/64 64x64/128 128x64/64 128x64/128 128x128/64 128x128/128"
{
echo "## fp32 train bs 8: forward with BN statistics"
timeout 600 python scripts/bench_layer.py --variants "$V" \
  f32:stats:8,256,32,32,1024,1,1,0 f32:stats:8,1024,32,32,256,1,1,0 f32:stats:8,512,16,16,2048,1,1,0 f32:stats:8,2048,16,16,512,1,1,0 \
  f32:stats:8,256,32,32,256,3,1,1 f32:stats:8,512,16,16,512,3,1,1 f32:stats:8,128,64,64,512,1,1,0 f32:stats:8,512,64,64,128,1,1,0
echo "## fp32 train bs 8: data gradients into a BatchNorm"
timeout 600 python scripts/bench_layer.py --variants "$V" \
  f32:bwd+res:8,1024,32,32,256,1,1,0 f32:bwd:8,256,32,32,1024,1,1,0 f32:bwd+res:8,2048,16,16,512,1,1,0 f32:bwd:8,512,16,16,2048,1,1,0 \
  f32:bwd:8,256,32,32,256,3,1,1 f32:bwd:8,512,16,16,512,3,1,1 f32:bwd+res:8,512,64,64,128,1,1,0 f32:bwd:8,128,64,64,512,1,1,0
} 2>&1 | grep -v Warn | tee $OUT/sweep_f32_small.txt
echo "=== done ($(date +%T))"
