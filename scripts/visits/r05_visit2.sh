#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v2; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou"
timeout 300 $B --no-parity --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3 > $OUT/train.log 2>&1; echo "exit $?"; tail -12 $OUT/train.log | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_train_ops.py -m gpu -q --timeout 600 -k "epilogue_wave or fused_statistics or halo_3x3 or batchnorm or decoder_wgrad_fp32 or relu_mask_as_bits" > $OUT/pytest_new.log 2>&1; echo "exit $?"; tail -8 $OUT/pytest_new.log | cut -c1-300
