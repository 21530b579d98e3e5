#!/bin/bash
# Tile / row-size sweep over representative layers of the two benchmark legs (scripts/bench_layer.py): the table the
# pick_tile / pick_rowb heuristics are fitted to.
V4="auto 128x128/64 128x128/128 128x64/64 128x64/128 256x128/64 256x128/128"
V8="$V4 256x256/128"
echo "## fp32 predict, bs 16 (encoder 1x1 group, 3x3, decoder phase)"
timeout 600 python scripts/bench_layer.py --variants "$V4" \
  f32:conv+res:16,64,128,128,256,1,1,0 f32:conv:16,256,128,128,64,1,1,0 f32:conv+res:16,128,64,64,512,1,1,0 f32:conv:16,512,64,64,128,1,1,0 \
  f32:conv+res:16,256,32,32,1024,1,1,0 f32:conv:16,1024,32,32,256,1,1,0 f32:conv+res:16,512,16,16,2048,1,1,0 f32:conv:16,2048,16,16,512,1,1,0 \
  f32:conv:16,64,128,128,64,3,1,1 f32:conv:16,128,64,64,128,3,1,1 f32:conv:16,256,32,32,256,3,1,1 f32:conv:16,512,16,16,512,3,1,1 \
  f32:phase:16,256+64,128,128,128 f32:phase:16,1024+256,32,32,256 f32:conv:16,32,512,512,32,3,1,1
echo "## bf16 train, bs 32: forward with BN statistics"
timeout 600 python scripts/bench_layer.py --variants "$V4" \
  bf16:stats:32,64,128,128,256,1,1,0 bf16:stats:32,256,128,128,64,1,1,0 bf16:stats:32,128,64,64,512,1,1,0 bf16:stats:32,512,64,64,128,1,1,0 \
  bf16:stats:32,256,32,32,1024,1,1,0 bf16:stats:32,1024,32,32,256,1,1,0 bf16:stats:32,64,128,128,64,3,1,1 bf16:stats:32,128,64,64,128,3,1,1 \
  bf16:stats:32,256,32,32,256,3,1,1 bf16:stats:32,512,16,16,512,3,1,1
echo "## bf16 train, bs 32: data gradients into a BatchNorm"
timeout 600 python scripts/bench_layer.py --variants "$V4" \
  bf16:bwd+res:32,256,128,128,64,1,1,0 bf16:bwd:32,64,128,128,256,1,1,0 bf16:bwd+res:32,512,64,64,128,1,1,0 bf16:bwd:32,128,64,64,512,1,1,0 \
  bf16:bwd:32,64,128,128,64,3,1,1 bf16:bwd:32,128,64,64,128,3,1,1 bf16:bwd:32,256,32,32,256,3,1,1 bf16:bwd:32,512,16,16,512,3,1,1
echo "## bf16 train, bs 32: decoder (phase form, its 4x4/s2 data gradient), dec5"
timeout 600 python scripts/bench_layer.py --variants "$V8" \
  bf16:phase:32,2048+256,16,16,256 bf16:phase:32,1024+256,32,32,256 bf16:phase:32,512+256,64,64,64 bf16:phase:32,256+64,128,128,128 bf16:phase:32,128,256,256,32 \
  bf16:dg4:32,256,16,16,2304 bf16:dg4:32,256,32,32,1280 bf16:dg4:32,64,64,64,768 bf16:dg4:32,128,128,128,320 bf16:dg4:32,32,256,256,128 \
  bf16:conv:32,32,512,512,32,3,1,1
