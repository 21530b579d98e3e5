set -x
OUT=gpurun_out/r04d; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -8 $OUT/pytest_gpu.log | cut -c1-250
B="python bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity"
for H in 1 0 1 0; do
  RS_CONV_HALO=$H timeout 600 $B --phase train --dtype bf16 --batch 32 --steps 20 --warmup 3 --full-json $OUT/train_halo$H.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('halo=$H train', d['value'], d['ms_per_step'], d['step_ms'])"
done
