#!/bin/bash
# Round-5 box visit 34: does the caching allocator grow inside the timed steps when the host runs far ahead?  Train leg alone, 30 steps.
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v34; mkdir -p $OUT
for RA in 0 2 0 2 0 2; do
  ROBOSAT_BENCH_RUNAHEAD=$RA timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --train-steps 30 --no-cpu-baseline --no-miou --no-extra-legs --no-parity --full-json $OUT/full.json > $OUT/bench.log 2>&1
  python - $OUT/full.json $RA <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); t = d["train"]
print("runahead", sys.argv[2], "predict", d["value"], "allocs", d["prewarm"]["device_allocs_in_timed_steps"], "| train", t["value"], "median", t["step_ms"]["median"], "max", t["step_ms"]["max"],
      "allocs in timed steps", t["device_allocs_in_timed_steps"], [round(v, 1) for v in t["step_ms"]["all"][:8]])
PY
done 2>&1 | tee $OUT/runahead.txt
