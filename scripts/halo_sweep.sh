#!/bin/bash
# A/B of the halo-once forms against the implicit-GEMM tiles on the layers of the bf16 train step they can run
# (scripts/bench_layer.py; "auto" = what the dispatcher picks, "halo/0" = the halo form forced).
V="auto halo/0 halo/64 128x128/128 256x128/128 256x256/128"
echo "## bf16 bs 32: Bottleneck.conv2 forward with BN statistics (layer1..3)"
timeout 600 python scripts/bench_layer.py --variants "$V" \
  bf16:stats:32,64,128,128,64,3,1,1 bf16:stats:32,128,64,64,128,3,1,1 bf16:stats:32,256,32,32,256,3,1,1
echo "## bf16 bs 32: Bottleneck.conv2 data gradient into a BatchNorm"
timeout 600 python scripts/bench_layer.py --variants "$V" \
  bf16:bwd:32,64,128,128,64,3,1,1 bf16:bwd:32,128,64,64,128,3,1,1 bf16:bwd:32,256,32,32,256,3,1,1
echo "## bf16 bs 32: DecoderBlock phase form (dec1, dec2, dec3) and its 4x4/s2 data gradient"
timeout 900 python scripts/bench_layer.py --variants "$V" \
  bf16:phase:32,1024+256,32,32,256 bf16:phase:32,512+256,64,64,64 bf16:phase:32,256+64,128,128,128 \
  bf16:dg4:32,256,32,32,1280 bf16:dg4:32,64,64,64,768 bf16:dg4:32,128,128,128,320
