OUT=gpurun_out/r04f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -x -k "fused_bn_finalize" --timeout 600 2>&1 | grep -v amdgpu.ids | tail -25 | cut -c1-250
timeout 1200 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_bf16.py tests/test_gpu_unet.py -m gpu -q -x --timeout 600 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-250
B="python bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity"
for F in 1 0 1 0; do
  ROBOSAT_BN_FUSED_FINALIZE=$F timeout 600 $B --phase train --dtype bf16 --batch 32 --steps 20 --warmup 3 --full-json $OUT/train_fin$F.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused_finalize=$F train', d['value'], d['ms_per_step'], d['step_ms'])"
done
