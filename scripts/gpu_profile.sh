#!/bin/bash
# Evidence run for profiles/: rocprofv3 kernel-trace stats of the default bench command + the two HBM-traffic counter
# passes (FETCH_SIZE, WRITE_SIZE each in its own --pmc run, kernel-trace only: MI355X_MICROARCH.md "HBM" / "PMC slots"),
# on the predict bench and on a short fp32 train bench whose stem bn_apply kernel (pure 16 B/lane streaming of a known
# tensor) calibrates the counters' units.
# usage: scripts/gpu_profile.sh TAG
TAG=${1:-r01}
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/profile_$TAG; mkdir -p $OUT; cd /tmp
B="python $REPO/bench.py --no-cpu-baseline --no-train-leg"
echo "== kernel trace: predict (default bench config)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/predict_trace -o p -- $B --steps 5 --warmup 2 > $OUT/predict_trace.log 2>&1; echo "exit $?"
echo "== kernel trace: train bf16 bs32"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_bf16_trace -o p -- $B --phase train --dtype bf16 --batch 32 --steps 3 --warmup 1 > $OUT/train_bf16_trace.log 2>&1; echo "exit $?"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C: predict"
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/predict_pmc_$C -o p -- $B --steps 1 --warmup 1 > $OUT/predict_pmc_$C.log 2>&1; echo "exit $?"
  echo "== pmc $C: train fp32 bs8 (calibration)"
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/trainf32_pmc_$C -o p -- $B --phase train --batch 8 --steps 1 --warmup 1 > $OUT/trainf32_pmc_$C.log 2>&1; echo "exit $?"
  echo "== pmc $C: train bf16 bs32"
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/trainbf16_pmc_$C -o p -- $B --phase train --dtype bf16 --batch 32 --steps 1 --warmup 1 > $OUT/trainbf16_pmc_$C.log 2>&1; echo "exit $?"
done
cd $REPO
# keep the merge small: per-dispatch traces are large
find $OUT -name "*kernel_trace.csv" -size +8M -delete
du -sh $OUT
