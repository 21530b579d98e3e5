#!/bin/bash
# Round-5 box visit 35: wgrad_phase4 off by default: the eager-vs-graphed step test six times, then the bf16 / race-screen / tile tests.
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v35; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_train_step.py -m gpu -q --timeout 300 -k "graphed_train_step" --count 1 -p no:cacheprovider > $OUT/g0.log 2>&1
for i in 1 2 3 4 5 6; do
  timeout 200 python -m pytest tests/test_gpu_train_step.py -m gpu -q --timeout 300 -k "graphed_train_step" > $OUT/g$i.log 2>&1; echo "run $i exit $? $(tail -1 $OUT/g$i.log | cut -c1-80)"
done
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_race_screen.py -m gpu -q -x --timeout 300 > $OUT/bf16.log 2>&1; echo "bf16 exit $? $(tail -1 $OUT/bf16.log | cut -c1-80)"
