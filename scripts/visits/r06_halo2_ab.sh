#!/bin/bash
# Round 6: the halo forms on the 256-pixel patch with 32-channel chunks, two blocks per CU (knob conv_halo512 = 2), against the rules
export TMPDIR=/tmp
for ROUND in 1 2; do
  for K in -1 2 0; do
    echo "## RS_CONV_HALO512=$K (round $ROUND)"
    RS_CONV_HALO512=$K timeout 300 python scripts/bench_layer.py --iters 30 --variants "auto" \
      bf16:stats:32,128,64,64,128,3,1,1 bf16:stats:32,256,32,32,256,3,1,1 bf16:bwd:32,128,64,64,128,3,1,1 bf16:bwd:32,256,32,32,256,3,1,1 \
      bf16:phase:32,256+64,128,128,128 bf16:dg4:32,256,32,32,1280 bf16:dg4:32,128,128,128,320 2>&1 | grep -v amdgpu.ids
  done
done
RS_CONV_HALO512=2 python -m pytest tests/test_gpu_tiles.py -m gpu -q -k "halo" 2>&1 | tail -2
