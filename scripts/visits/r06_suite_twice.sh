#!/bin/bash
# the driver's exact GPU command, twice in one visit (two processes, the race screen at its alphabetical place in the order)
export TMPDIR=/tmp
OUT=gpurun_out/r6suite; mkdir -p $OUT
for i in 1 2; do
  timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/run$i.log 2>&1; echo "run $i exit $? $(tail -1 $OUT/run$i.log | cut -c1-120)"
  grep -E "positive control|INCONCLUSIVE" $OUT/run$i.log | head -2
done
