#!/bin/bash
# Round evidence, part 2 (after scripts/gpu_profile.sh + scripts/pmc_traffic.py refreshed profiles/pmc_traffic.json):
# the full GPU test log, the default bench line, the train bench line and the per-layer tables.
# usage: scripts/gpu_evidence.sh TAG
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/evidence_$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 3 --layers-json $OUT/layers_predict.json > $OUT/bench_default.log 2>&1; echo "bench exit $?"
tail -1 $OUT/bench_default.log > $OUT/bench_default.json; cut -c1-600 $OUT/bench_default.json
timeout 600 python bench.py --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --layers-json $OUT/layers_train.json > $OUT/bench_train.log 2>&1; echo "train bench exit $?"
tail -1 $OUT/bench_train.log > $OUT/bench_train_bf16_bs32.json; cut -c1-400 $OUT/bench_train_bf16_bs32.json
# the non-headline configurations (one line each)
{
  for A in "--dtype bf16 --no-train-leg --steps 20" "--phase train --batch 8 --steps 5 --warmup 2" "--size 1024 --batch 8 --no-train-leg --steps 10" "--phase train --dtype bf16 --batch 32 --classes 4 --steps 10" "--size 576 --batch 16 --no-train-leg --steps 10"; do
    timeout 600 python bench.py --no-cpu-baseline $A 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '->', d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['dtype'], '|', d['config']['workload'])" "$A"
  done
} > $OUT/bench_others.txt 2>&1
cat $OUT/bench_others.txt
