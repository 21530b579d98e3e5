#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v3; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou"
timeout 300 $B --no-parity --phase train --dtype fp32 --batch 8 --steps 5 --warmup 2 --layers-json $OUT/layers_train_fp32.json --full-json $OUT/train_fp32_full.json > $OUT/train_fp32.log 2>&1; echo "exit $?"; tail -1 $OUT/train_fp32.log | cut -c1-400
timeout 300 $B --no-parity --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3 --layers-json $OUT/layers_train_bf16.json --full-json $OUT/train_bf16_full.json > $OUT/train_bf16.log 2>&1; echo "exit $?"; tail -1 $OUT/train_bf16.log | cut -c1-400
for CH in 3 4 3; do
  timeout 300 $B --no-parity --phase train --dtype bf16 --batch 32 --classes 4 --channels $CH --steps 10 --warmup 3 --full-json $OUT/train_c4_ch$CH.json 2>$OUT/bands_err_$CH.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('channels $CH', d['value'], d['ms_per_step'], d['step_ms'])"
done | tee $OUT/bands_ab.txt
