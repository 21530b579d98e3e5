#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v20; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3"
timeout 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 300 -k "wgrad" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log | cut -c1-300
run() { env "$1" timeout 200 $B --full-json $OUT/full_$2.json 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'])"; }
{
run X=warm w
run RS_WGRAD_PHASE4=0 p0
run RS_WGRAD_PHASE4=1 p1
run RS_WGRAD_PHASE4=0 p0b
run RS_WGRAD_PHASE4=1 p1b
run RS_WGRAD_BLOCKS_PHASE4=256 b256
run RS_WGRAD_BLOCKS_PHASE4=384 b384
run RS_WGRAD_BLOCKS_PHASE4=768 b768
} | tee $OUT/phase4_ab.txt
python - <<'PY'
import json
for t in ('p0','p1'):
    d=json.load(open('gpurun_out/v20/full_%s.json'%t))
    for n,v in d['roofline']['per_kernel'].items():
        if 'wgrad_bf16<phase' in n: print(t, n, v)
PY
