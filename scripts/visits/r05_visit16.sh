#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v16; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou"
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -m gpu -q --timeout 300 -k "stem or probs_match or unet" > $OUT/pytest_stem.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_stem.log | cut -c1-300
for E in 0 1 0 1; do
  RS_STEM_DMA=$E timeout 200 $B --no-train-leg --steps 20 --warmup 5 --full-json $OUT/predict_stem$E.json 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('RS_STEM_DMA=$E', d['value'], d['ms_per_step'], d['step_ms']['median'], d['parity'])"
done | tee $OUT/stem_ab.txt
python - <<'PY'
import json
for E in (0,1):
    d=json.load(open('gpurun_out/v16/predict_stem%d.json'%E))
    for n,v in d['roofline']['per_kernel'].items():
        if 'stem' in n: print(E, n, v)
PY
