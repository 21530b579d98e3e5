#!/bin/bash
# A/B of variants of the two fp32 Winograd kernels on the predict pass (env knobs of conv_wino33_f32.hip / conv_wino_f32.hip):
# (The knobs RS_WINO33_WPC / RS_WINO33_VAR / RS_WINO_VAR exist only in commit 0489e6e, which the next commit reverts: check that
#  commit out to repeat the measurement; results are in profiles/r04/wino_variants_r04.txt.)
# parity tests of the kernels under the variant first, then the predict bench with each setting.
OUT=gpurun_out/${1:-wpc}; mkdir -p $OUT
timeout 60 python -c "import torch; x = torch.arange(1 << 20, device='cuda:0', dtype=torch.float32); assert float((x * 2).sum().cpu()) == float((1 << 20) * ((1 << 20) - 1))" || { echo "GPU sanity failed"; exit 3; }
RS_WINO33_VAR=4 timeout 400 python -m pytest tests/test_gpu_tiles.py -m gpu -q -k "winograd_3x3" --timeout 300 > $OUT/pytest.log 2>&1; echo "pytest (variant on) exit $?"; tail -4 $OUT/pytest.log | cut -c1-300
B="python bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-train-leg --steps 20 --warmup 5"
run() {  # tag, env...
  local T=$1; shift
  env "$@" timeout 200 $B $NP --layers-json $OUT/layers_$T.json --full-json $OUT/full_$T.json > $OUT/bench_$T.log 2>&1; echo "bench $T exit $?"
  tail -1 $OUT/bench_$T.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['value'], d['unit'], d['ms_per_step'], 'ms', d.get('parity'))" $T
  python - $OUT/layers_$T.json <<'P'
import json,sys,collections
agg=collections.defaultdict(float)
for r in json.load(open(sys.argv[1])): agg[r['kernel']]+=r['ms']
print('   ' + '  '.join('%s %.1f' % (k.replace('conv_wino_f32',''), v*1e3) for k,v in sorted(agg.items(), key=lambda x:-x[1]) if 'wino' in k))
P
}
NP=""
run base A=0
run w33two RS_WINO33_VAR=4
NP="--no-parity"
run w33ko6 RS_WINO33_VAR=6
run wko2 RS_WINO_VAR=2
run wko4 RS_WINO_VAR=4
run wko6 RS_WINO_VAR=6
