#!/bin/bash
# full GPU suite + the driver's bench command on the current tree
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v13; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -rP --maxfail 20 --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_gpu.log | cut -c1-300
grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --layers-json $OUT/layers_predict.json --full-json $OUT/bench_default_full.json > $OUT/bench_default.log 2>&1; echo "bench exit $?"
tail -1 $OUT/bench_default.log > $OUT/bench_default.json; python scripts/bench_brief.py $OUT/bench_default.json
