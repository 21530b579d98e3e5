#!/bin/bash
# Do one-off stalls in the timed legs go away with the settle phase?  N runs of the driver's command each way:
#   old = round 2's behaviour (no settle, torch.cuda.empty_cache() between legs), new = this round's.
# Prints every leg's min / median / max and which step was slowest.   bash scripts/stall_ab.sh OUTDIR [N]
OUT=$1; N=${2:-4}; mkdir -p $OUT
for i in $(seq 1 $N); do
  for MODE in old new; do
    if [ $MODE = old ]; then export ROBOSAT_BENCH_PREWARM=0 ROBOSAT_BENCH_EMPTY_CACHE=1; else unset ROBOSAT_BENCH_PREWARM ROBOSAT_BENCH_EMPTY_CACHE; fi
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-miou --no-parity > $OUT/ab_${MODE}_$i.log 2>&1
    tail -1 $OUT/ab_${MODE}_$i.log > $OUT/ab_${MODE}_$i.json
    echo "== $MODE run $i"; python scripts/bench_brief.py $OUT/ab_${MODE}_$i.json | cut -c1-150
  done
done
