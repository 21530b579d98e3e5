#!/bin/bash
# A/B of the weight-gradient split target (RS_WGRAD_BLOCKS) on the bf16 train step: step time + per-kernel wgrad sums
for B in "256 1536" "128 1536" "192 1536" "256 1024" "256 768" "192 768" "256 1536" "320 1536"; do
  set -- $B
  echo "== RS_WGRAD_BLOCKS=$1 RS_WGRAD_BLOCKS_PHASE=$2"
  RS_WGRAD_BLOCKS=$1 RS_WGRAD_BLOCKS_PHASE=$2 timeout 300 python bench.py --no-cpu-baseline --no-parity --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['per_kernel']
print(d['ms_per_step'], {k:(v['ms'],v['launches']) for k,v in r.items() if 'wgrad' in k})"
done
