#!/bin/bash
# Round-5 box visit 23: kernel traces of the fp32 train step (bs 8), two streams and one, for a per-kernel view of the leg
# VERDICT r4 asks to bring under 25 ms.
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v23; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity"
if ! timeout 90 python -c "import torch; x = torch.arange(1 << 20, device='cuda:0', dtype=torch.float32); assert float((x * 2).sum().cpu()) == float((1 << 20) * ((1 << 20) - 1))" > $OUT/sanity.log 2>&1; then
  echo "=== GPU sanity check FAILED"; tail -3 $OUT/sanity.log; exit 3
fi
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_f32 -o p -- $B --phase train --dtype fp32 --batch 8 --steps 5 --warmup 2 > $OUT/trace_f32.log 2>&1; echo "exit $?"
ROBOSAT_WGRAD_STREAM=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_f32_serial -o p -- $B --phase train --dtype fp32 --batch 8 --steps 5 --warmup 2 > $OUT/trace_f32_serial.log 2>&1; echo "exit $?"
cd $REPO
for i in 1 2; do
  timeout 200 $B --phase train --dtype fp32 --batch 8 --steps 10 --warmup 3 2>$OUT/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 bs8', d['value'], d['ms_per_step'], d['step_ms'])"
done | tee $OUT/f32_step.txt
find $OUT -name "*.db" -delete; find $OUT -name "*agent_info*" -delete
echo "=== done ($(date +%T))"
