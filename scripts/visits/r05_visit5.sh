#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v5; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou"
for T in 1024 512 2048 256; do
  RS_WGRAD_F32_BLOCKS=$T timeout 300 $B --no-parity --phase train --dtype fp32 --batch 8 --steps 5 --warmup 2 --layers-json $OUT/layers_$T.json --full-json $OUT/full_$T.json 2>$OUT/err_$T.log | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('RS_WGRAD_F32_BLOCKS=$T', d['value'], d['ms_per_step'], d['step_ms']['median'], r['kernel'], r.get('frac'))"
done | tee $OUT/blocks_ab.txt
