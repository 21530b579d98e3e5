#!/bin/bash
# fp32 LDS-DMA weight gradient: first run -- parity tests, then the fp32 train leg A/B (knob 0 / default) with per-kernel tables
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v4; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou"
timeout 600 python -m pytest tests/test_gpu_train_ops.py -m gpu -q --timeout 300 -k "wgrad or decoder" > $OUT/pytest_wgrad.log 2>&1; echo "pytest exit $?"; tail -12 $OUT/pytest_wgrad.log | cut -c1-300
for D in 0 1 0 1; do
  RS_WGRAD_F32_DMA=$D timeout 300 $B --no-parity --phase train --dtype fp32 --batch 8 --steps 5 --warmup 2 --full-json $OUT/train_fp32_dma$D.json 2>$OUT/err_$D.log | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('RS_WGRAD_F32_DMA=$D', d['value'], d['ms_per_step'], d['step_ms'], r['kernel'], r.get('frac'))"
done | tee $OUT/dma_ab.txt
python - <<'PY'
import json
for D in (0,1):
    d=json.load(open('gpurun_out/v4/train_fp32_dma%d.json'%D))
    pk=d['roofline']['per_kernel']
    for n,v in pk.items():
        if 'wgrad' in n: print(D, n, v)
PY
