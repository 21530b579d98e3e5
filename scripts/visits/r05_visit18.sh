#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v18; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou"
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train_step.py -m gpu -q --timeout 600 -k "weight_prep or golden or trajectory or optimizer or graph" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.log | cut -c1-300
for E in 1 2 3; do
  timeout 300 $B --no-parity --phase train --dtype fp32 --batch 8 --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fp32 train', d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'])"
done | tee $OUT/fp32_train.txt
