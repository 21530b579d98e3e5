#!/bin/bash
# Round-5 box visit 25: ring-of-three weight-gradient kernel with fewer, longer blocks (the ring keeps a lone block's DMA path busy).
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v25; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity --phase train --dtype bf16 --batch 32 --steps 20 --warmup 3"
run() { env "$@" timeout 200 $B 2>$OUT/err.log | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'])"; }
{
run X=warm
for i in 1 2; do
run RS_WGRAD_RING=2
run RS_WGRAD_RING=3 RS_WGRAD_BLOCKS=128
run RS_WGRAD_RING=3 RS_WGRAD_BLOCKS=96
run RS_WGRAD_RING=3 RS_WGRAD_BLOCKS=64
run RS_WGRAD_RING=2 RS_WGRAD_BLOCKS=128
done
} | tee $OUT/ring_blocks.txt
echo "=== done ($(date +%T))"
