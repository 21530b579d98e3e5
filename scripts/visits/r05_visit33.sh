#!/bin/bash
# Round-5 box visit 33: the default bench command twice (no CPU baseline / mIoU: the timed legs are what is looked at): first timed steps of every leg.
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v33; mkdir -p $OUT
for i in 1 2; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-miou --full-json $OUT/full$i.json > $OUT/bench$i.log 2>&1; echo "exit $?"
  python - $OUT/full$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("predict", d["value"], d["prewarm"]["settle_steps"], d["step_ms"]["all"][:6])
t = d["train"]; print("train  ", t["value"], t["settle_steps"], t["step_ms"]["median"], t["step_ms"]["all"][:8])
for k, v in d["legs"].items(): print(k, v["value"], v["settle_steps"], v["step_ms"]["all"][:4])
PY
done 2>&1 | tee $OUT/first_steps.txt
