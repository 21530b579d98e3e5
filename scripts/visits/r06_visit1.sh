#!/bin/bash
# Round-6 box visit 1: the race screens with their positive control, the bf16 / train-step suites with the rings back on by default,
# the eager-vs-graphed bisect script (20 runs at the defaults), and the bf16 step A/B rings on / off.
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/r6v5; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_race_screen.py -m gpu -q -rP --timeout 600 > $OUT/race.log 2>&1; echo "race screen exit $? $(tail -1 $OUT/race.log | cut -c1-100)"; grep -E "positive control|INCONCLUSIVE" $OUT/race.log | head -3
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train_step.py -m gpu -q --timeout 600 > $OUT/bf16.log 2>&1; echo "bf16 + train step exit $? $(tail -1 $OUT/bf16.log | cut -c1-100)"
timeout 600 python scripts/flaky_graph_step.py 10 1 > $OUT/flaky_graph.log 2>&1; echo "flaky exit $?"; tail -3 $OUT/flaky_graph.log | cut -c1-300
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity --phase train --dtype bf16 --batch 32 --steps 30 --warmup 3"
for i in 1 2 3; do
  for S in "X=default" "RS_WGRAD_RING=2 RS_WGRAD_PHASE4=0 RS_WGRAD_BLOCKS=192"; do
    env $S timeout 200 $B --full-json $OUT/ab.json > $OUT/ab.log 2>&1
    python - "$S" $OUT/ab.log <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    t = d.get("train", d)
    sm = t.get("step_ms", {})
    print("  ", sys.argv[1], "| tiles/s", t.get("value"), "ms/step", t.get("ms_per_step"), "median", sm.get("median"), "min", sm.get("min"), flush=True)
except Exception as e:
    print("  ", sys.argv[1], "bench line unreadable:", e)
PY
  done
done
