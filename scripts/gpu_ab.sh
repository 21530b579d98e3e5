#!/bin/bash
# Quick A/B on one box: touched parity tests, then the bf16 bs-32 train bench under env-knob variants.
# usage: scripts/gpu_ab.sh TAG "ENV1=.. ENV2=.." "ENVb=.." ...   (first variant "" = defaults)
TAG=${1:-ab}; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train_ops.py tests/test_gpu_bf16.py -m gpu -q -x --timeout 300 > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest exit $?"; tail -4 gpurun_out/pytest_$TAG.log
i=0
for V in "" "$@"; do
  echo "== variant $i: [$V]"
  env $V timeout 600 python bench.py --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  i=$((i+1))
done
