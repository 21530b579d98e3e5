OUT=gpurun_out/r04g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x -k "decoder or conv_backward or wgrad" --timeout 600 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_configs.py -m gpu -q -x -k "fp32 or cfg5_four or golden or step" --timeout 600 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-250
B="python bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity"
for F in phase direct phase direct; do
  if [ $F = direct ]; then export RS_WGRAD_F32_PHASE=0; else unset RS_WGRAD_F32_PHASE; fi
  timeout 600 $B --phase train --dtype fp32 --batch 8 --steps 10 --warmup 3 --full-json $OUT/train_fp32_$F.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$F fp32 train bs8', d['value'], d['ms_per_step'], d['step_ms'])"
done
