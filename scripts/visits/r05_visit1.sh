#!/bin/bash
# Round-5 box visit 1: (a) consecutive predict batches on alternating streams, (b) the K <= 64 rule of conv1x1_ew_f32 on the
# predict pass, (c) this round's new / touched tests, (d) the 3-band vs 4-band 4-class train legs with every step time,
# (e) LAST (first run of a kernel written without a GPU: a hang must not cost the rest) conv1x1_ew_bf16 parity + A/B.
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/v1; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou"
if ! timeout 90 python -c "import torch; x = torch.arange(1 << 20, device='cuda:0', dtype=torch.float32); assert float((x * 2).sum().cpu()) == float((1 << 20) * ((1 << 20) - 1))" > $OUT/sanity.log 2>&1; then
  echo "=== GPU sanity check FAILED"; tail -3 $OUT/sanity.log; exit 3
fi
echo "=== two-stream predict ($(date +%T))"
timeout 300 python scripts/two_stream_predict.py --split 2 > $OUT/two_stream_fp32.txt 2>&1; echo "exit $?"; cat $OUT/two_stream_fp32.txt | grep -v Warn
timeout 200 python scripts/two_stream_predict.py --dtype bf16 --streams 1 2 1 > $OUT/two_stream_bf16.txt 2>&1; echo "exit $?"; cat $OUT/two_stream_bf16.txt | grep -v Warn
timeout 200 python scripts/two_stream_predict.py --size 1024 --batch 8 --steps 10 --streams 1 2 1 > $OUT/two_stream_1024.txt 2>&1; echo "exit $?"; cat $OUT/two_stream_1024.txt | grep -v Warn
echo "=== predict pass, conv1x1_ew rule off / on ($(date +%T))"
for E in 0 -1 0 -1; do
  RS_CONV1X1_EW=$E timeout 200 $B --no-train-leg --no-parity --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RS_CONV1X1_EW=$E', d['value'], d['ms_per_step'], d['step_ms'])"
done | tee $OUT/ew_rule_ab.txt
echo "=== new tests ($(date +%T))"
timeout 900 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_train_ops.py -m gpu -q -x --timeout 600 -k "epilogue_wave or fused_statistics or halo_3x3 or batchnorm or decoder_wgrad_fp32 or relu_mask_as_bits" > $OUT/pytest_new.log 2>&1; echo "exit $?"; tail -5 $OUT/pytest_new.log | cut -c1-300
echo "=== 3-band vs 4-band, 4 classes ($(date +%T))"
for CH in 3 4 3; do
  timeout 300 $B --no-parity --phase train --dtype bf16 --batch 32 --classes 4 --channels $CH --steps 10 --warmup 3 --full-json $OUT/train_c4_ch$CH.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('channels $CH', d['value'], d['ms_per_step'], d['step_ms'])"
done | tee $OUT/bands_ab.txt
echo "=== conv1x1_ew_bf16: first run ($(date +%T))"
timeout 150 python scripts/ew_bf16_check.py > $OUT/ew_bf16_check.txt 2>&1; RC=$?; echo "exit $RC"; grep -v Warn $OUT/ew_bf16_check.txt | tail -30
if grep -q "PARITY OK" $OUT/ew_bf16_check.txt; then
  for E in 0 1 0 1; do
    RS_CONV1X1_EW_BF16=$E timeout 200 $B --no-parity --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RS_CONV1X1_EW_BF16=$E', d['value'], d['ms_per_step'], d['step_ms'])"
  done | tee $OUT/ew_bf16_step_ab.txt
fi
echo "=== done ($(date +%T))"
