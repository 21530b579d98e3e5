#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v14; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3"
for V in "256 1536" "192 1536" "128 1536" "384 1536" "256 1024" "256 2048" "256 1536"; do
  set -- $V
  RS_WGRAD_BLOCKS=$1 RS_WGRAD_BLOCKS_PHASE=$2 timeout 200 $B 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('WGRAD_BLOCKS=$1 PHASE=$2', d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'])"
done | tee $OUT/wgrad_blocks.txt
