#!/bin/bash
# Round-5 box visit 27: ring 3 @ 96 as the default: the bf16 / train-op / tile-coverage tests, then default vs the former setting
# on the bf16 step and on the 4-band 4-class leg.
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v27; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train_ops.py tests/test_gpu_tiles.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log | cut -c1-300
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity --phase train --dtype bf16 --batch 32 --steps 20 --warmup 3"
run() { env "$@" timeout 200 $B $EXTRA 2>$OUT/err.log | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$* $EXTRA', d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'])"; }
{
EXTRA="" run X=warm
for i in 1 2; do EXTRA="" run X=default; EXTRA="" run RS_WGRAD_RING=2 RS_WGRAD_BLOCKS=192; done
for i in 1 2; do EXTRA="--classes 4 --channels 4" run X=default; EXTRA="--classes 4 --channels 4" run RS_WGRAD_RING=2 RS_WGRAD_BLOCKS=192; done
} | tee $OUT/default_ab.txt
echo "=== done ($(date +%T))"
