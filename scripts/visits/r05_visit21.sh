#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v21; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity --phase train --dtype bf16 --batch 32 --steps 20 --warmup 3"
run() { env "$1" timeout 200 $B 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'])"; }
{
run X=warm
for i in 1 2 3; do run RS_WGRAD_PHASE4=0; run RS_WGRAD_PHASE4=1; done
run RS_WGRAD_BLOCKS_PHASE4=192
run RS_WGRAD_BLOCKS_PHASE4=128
run RS_WGRAD_BLOCKS_PHASE4=320
} | tee $OUT/phase4_ab2.txt
