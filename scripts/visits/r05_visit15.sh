#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v15; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3"
run() { env "$@" timeout 200 $B 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'])"; }
{
run X=warmup
run X=base
run RS_WGRAD_BLOCKS=96
run RS_WGRAD_BLOCKS=64
run RS_CONV_HALO_MIN=128
run RS_CONV_HALO_MIN=256
run RS_CONV_MIN256=256
run RS_CONV_MIN256=512
run RS_CONV_HALO512=1
run RS_CONV_HALO512=0
run X=base
} | tee $OUT/knob_sweep.txt
