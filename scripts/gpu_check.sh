#!/bin/bash
# One gpurun call: GPU parity tests, smoke, a short bench, and a rocprofv3 kernel trace of the bench.
# usage: scripts/gpu_check.sh [tag] [pytest-args...]
TAG=${1:-r1}; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest -m gpu" 
timeout 900 python -m pytest tests -m gpu -q --timeout 300 "$@" > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest exit $?"; tail -25 gpurun_out/pytest_$TAG.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 2 --layers-json gpurun_out/layers_$TAG.json > gpurun_out/bench_$TAG.log 2>&1
echo "bench exit $?"; tail -3 gpurun_out/bench_$TAG.log
echo "== bench train"
timeout 900 python bench.py --phase train --batch 8 --steps 3 --warmup 1 --cpu-seconds 10 --layers-json gpurun_out/layers_train_$TAG.json > gpurun_out/bench_train_$TAG.log 2>&1
echo "bench train exit $?"; tail -3 gpurun_out/bench_train_$TAG.log
if [ -n "$ROCPROF" ]; then
echo "== rocprofv3 kernel trace (predict bench)"
REPO=$(pwd); cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o predict -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-leg > $REPO/gpurun_out/rocprof_$TAG.log 2>&1
echo "rocprof exit $?"; tail -2 $REPO/gpurun_out/rocprof_$TAG.log
find $REPO/gpurun_out/prof_$TAG -name "*stats*" | head
cd $REPO
fi
