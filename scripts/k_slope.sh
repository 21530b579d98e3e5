#!/bin/bash
# Fixed vs per-chunk cost of a 1x1 launch: the same pixel tile grid and cout count at K = 32 ... 1024 input channels, with the
# eval epilogue, the train-mode statistics epilogue and the data-gradient-into-BatchNorm epilogue (scripts/bench_layer.py).
# The intercept at K -> 0 is what a tile costs before its first MFMA and after its last (table, prologue DMA round trip,
# epilogue); the slope is the cost of a 32-channel (bf16, r64) chunk.  fp32 the same at K = 16 ... 512 (r64 = 16 channels).
timeout 60 python -c "import torch; x = torch.arange(1 << 20, device='cuda:0', dtype=torch.float32); assert float((x * 2).sum().cpu()) == float((1 << 20) * ((1 << 20) - 1))" || { echo "GPU sanity failed"; exit 3; }
S=""
for K in 32 64 128 256 512 1024; do S="$S bf16:stats:32,$K,32,32,1024,1,1,0 bf16:conv:32,$K,32,32,1024,1,1,0 bf16:bwd+res:32,$K,32,32,1024,1,1,0"; done
for K in 32 64 128 256 512; do S="$S bf16:stats:32,$K,64,64,512,1,1,0"; done
timeout 200 python scripts/bench_layer.py --iters 30 --variants "128x128/64" $S 2>/dev/null | cut -c1-170
S=""
for K in 16 32 64 128 256 512; do S="$S f32:conv+res:16,$K,32,32,1024,1,1,0 f32:conv+res:16,$K,128,128,256,1,1,0"; done
timeout 200 python scripts/bench_layer.py --iters 30 --variants "128x128/64 128x64/64" $S 2>/dev/null | cut -c1-170
