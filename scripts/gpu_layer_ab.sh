#!/bin/bash
# usage: scripts/gpu_layer_ab.sh "bench_layer args" "ENV=.." "ENV=.." ...
ARGS=$1; shift
for V in "" "$@"; do
  echo "== [$V]"
  env $V timeout 300 python scripts/bench_layer.py $ARGS 2>&1 | tail -12
done
