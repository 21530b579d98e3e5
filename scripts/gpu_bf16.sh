#!/bin/bash
# One gpurun call for the bf16 path: hardware probes, bf16 parity tests, fp32 regression tests, bf16 benches.
TAG=${1:-bf}; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== probes"
timeout 60 scripts/probes/probe_glds.bin 2>&1 | tail -12
timeout 60 scripts/probes/probe_tr16.bin > gpurun_out/probe_tr16_$TAG.log 2>&1; echo "tr16 exit $?"
echo "== pytest bf16"
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 300 -x "$@" > gpurun_out/pytest_bf16_$TAG.log 2>&1
echo "pytest bf16 exit $?"; tail -30 gpurun_out/pytest_bf16_$TAG.log
echo "== pytest fp32 regression"
timeout 900 python -m pytest tests -m gpu -q --timeout 300 --deselect tests/test_gpu_bf16.py > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest exit $?"; tail -5 gpurun_out/pytest_$TAG.log
echo "== bench train bf16 bs8"
timeout 600 python bench.py --phase train --dtype bf16 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --layers-json gpurun_out/layers_train_bf16_$TAG.json > gpurun_out/bench_train_bf16_$TAG.log 2>&1
echo "exit $?"; tail -2 gpurun_out/bench_train_bf16_$TAG.log
echo "== bench predict bf16 bs16"
timeout 600 python bench.py --dtype bf16 --steps 10 --warmup 2 --no-cpu-baseline --no-train-leg --layers-json gpurun_out/layers_bf16_$TAG.json > gpurun_out/bench_bf16_$TAG.log 2>&1
echo "exit $?"; tail -2 gpurun_out/bench_bf16_$TAG.log
