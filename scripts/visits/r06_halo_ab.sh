#!/bin/bash
# Round 6: A/B of two builds of the library on the halo-form layers of the bf16 bs-32 step (same box, alternating):
#   scripts/visits/r06_halo_ab.sh gpurun_in/lib_A.so gpurun_in/lib_B.so
export TMPDIR=/tmp
REPO=$(pwd); LIB=$REPO/robosat_amd/librobosat_hip.so
cp $LIB /tmp/lib_orig.so
for ROUND in 1 2; do
  for L in "$@"; do
    cp $REPO/$L $LIB
    echo "## $L (round $ROUND)"
    timeout 300 python scripts/bench_layer.py --iters 30 --variants "auto" \
      bf16:stats:32,128,64,64,128,3,1,1 bf16:stats:32,256,32,32,256,3,1,1 bf16:bwd:32,128,64,64,128,3,1,1 bf16:bwd:32,256,32,32,256,3,1,1 \
      bf16:phase:32,512+256,64,64,64 bf16:phase:32,256+64,128,128,128 bf16:dg4:32,256,32,32,1280 bf16:dg4:32,128,128,128,320 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/lib_orig.so $LIB
