#!/bin/bash
# Round-5 box visit 24: conv_wgrad_bf16 with a ring of three chunk buffers (knob wgrad_ring = 3): parity of every bf16 weight-
# gradient test with the ring on, then the bf16 step with the ring off / on, alternating, and two block targets with it on.
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v24; mkdir -p $OUT
if ! timeout 90 python -c "import torch; x = torch.arange(1 << 20, device='cuda:0', dtype=torch.float32); assert float((x * 2).sum().cpu()) == float((1 << 20) * ((1 << 20) - 1))" > $OUT/sanity.log 2>&1; then
  echo "=== GPU sanity check FAILED"; tail -3 $OUT/sanity.log; exit 3
fi
RS_WGRAD_RING=3 timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train_ops.py -m gpu -q -x --timeout 300 -k "wgrad" > $OUT/pytest_ring3.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_ring3.log | cut -c1-300
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity --phase train --dtype bf16 --batch 32 --steps 20 --warmup 3"
run() { env "$@" timeout 200 $B 2>$OUT/err.log | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'])"; }
{
run X=warm
for i in 1 2 3; do run RS_WGRAD_RING=2; run RS_WGRAD_RING=3; done
run RS_WGRAD_RING=3 RS_WGRAD_BLOCKS=128
run RS_WGRAD_RING=3 RS_WGRAD_BLOCKS=256
} | tee $OUT/ring_ab.txt
# per-launch view: the serial roofline pass of one leg each
for R in 2 3; do
  RS_WGRAD_RING=$R timeout 200 python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity --phase train --dtype bf16 --batch 32 --steps 5 --warmup 2 --full-json $OUT/full_ring$R.json > /dev/null 2>&1
done
echo "=== done ($(date +%T))"
