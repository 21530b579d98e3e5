#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v6; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou"
timeout 600 python -m pytest tests/test_gpu_train_ops.py -m gpu -q --timeout 300 -k "wgrad or decoder" > $OUT/pytest_wgrad.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_wgrad.log | cut -c1-300
for V in "16 2048" "32 2048" "16 4096" "16 1024" "16 2048"; do
  set -- $V
  RS_WGRAD_F32_PK=$1 RS_WGRAD_F32_BLOCKS=$2 timeout 300 $B --no-parity --phase train --dtype fp32 --batch 8 --steps 5 --warmup 2 --layers-json $OUT/layers_$1_$2.json --full-json $OUT/full_$1_$2.json 2>$OUT/err.log | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('PK=$1 BLOCKS=$2', d['value'], d['ms_per_step'], d['step_ms']['median'], r['kernel'], r.get('frac'))"
done | tee $OUT/pk_ab.txt
