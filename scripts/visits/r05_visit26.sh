#!/bin/bash
# Round-5 box visit 26: confirmation of ring 3 @ 96 blocks against the shipped ring 2 @ 192 (three alternations, 30 steps each).
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v26; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity --phase train --dtype bf16 --batch 32 --steps 30 --warmup 3"
run() { env "$@" timeout 200 $B 2>$OUT/err.log | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'])"; }
{
run X=warm
for i in 1 2 3; do
run RS_WGRAD_RING=2
run RS_WGRAD_RING=3 RS_WGRAD_BLOCKS=96
run RS_WGRAD_RING=3 RS_WGRAD_BLOCKS=80
run RS_WGRAD_RING=3 RS_WGRAD_BLOCKS=112
done
} | tee $OUT/ring_confirm.txt
echo "=== done ($(date +%T))"
