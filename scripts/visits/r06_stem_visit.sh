#!/bin/bash
# the fp32 stem on its own kernel (stem_f32.hip): parity first, then the predict leg and the layer table
export TMPDIR=/tmp
OUT=gpurun_out/r6stem; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "stem" > $OUT/tests_stem.log 2>&1; echo "stem tests exit $? $(tail -1 $OUT/tests_stem.log | cut -c1-150)"
grep -E "Error|error|assert" $OUT/tests_stem.log | head -10
echo skip-net-tests
B="python bench.py --no-cpu-baseline --no-extra-legs --no-miou"
for i in 1 2; do
  timeout 300 $B --no-train-leg --steps 30 --warmup 5 --layers-json $OUT/layers_predict_$i.json > $OUT/predict_$i.log 2>&1
  echo "predict run $i: $(tail -1 $OUT/predict_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("step_ms"), d.get("parity"))')"
done
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6stem/layers_predict_1.json'))
for r in d[:3]: print(r['kernel'], r['cin_cout_k_stride_ups_ho_wo'], round(r['ms']*1000,1), 'us', round(r['tflops'],1), 'TF')
print('sum ms', sum(r['ms'] for r in d))
PY
timeout 300 $B --phase train --dtype fp32 --batch 8 --steps 20 --warmup 5 --no-parity > $OUT/f32_train.log 2>&1
echo "fp32 train: $(tail -1 $OUT/f32_train.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("step_ms"))')"
