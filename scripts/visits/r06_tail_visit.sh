#!/bin/bash
# layer1's fused Bottleneck tail (bottleneck_tail_f32.hip): parity, then the predict pass with it on / off
export TMPDIR=/tmp
OUT=gpurun_out/r6tail; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -rP -k "tail or stem or wave" > $OUT/tests.log 2>&1; echo "tests exit $? $(tail -1 $OUT/tests.log | cut -c1-150)"; grep -E "fused vs unfused|Error|assert " $OUT/tests.log | head -5
B="python bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-train-leg"
for i in 1 2; do
  for K in 0 1; do
    ROBOSAT_TAIL_FUSE=$K timeout 300 $B --steps 30 --warmup 5 --layers-json $OUT/layers_${K}_$i.json > $OUT/predict_${K}_$i.log 2>&1
    echo "tail fuse=$K run $i: $(tail -1 $OUT/predict_${K}_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("step_ms"), d.get("parity"))')"
  done
done
python - <<'PY'
import json
for K in (0, 1):
    d = json.load(open('gpurun_out/r6tail/layers_%d_1.json' % K))
    print("fuse", K, "layers", len(d), "sum ms %.3f" % sum(r['ms'] for r in d))
    for r in d[1:14]: print("   ", r['kernel'], r['cin_cout_k_stride_ups_ho_wo'][:4], round(r['ms'] * 1000, 1), 'us')
PY
