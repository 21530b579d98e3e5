timeout 600 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_bf16.py tests/test_gpu_train_step.py -m gpu -q -x --timeout 300 2>&1 | tail -3
for V in "" ""; do echo "== [$V]"
env $V timeout 300 python bench.py --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
ROBOSAT_WGRAD_STREAM=0 timeout 300 python bench.py --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial', d['ms_per_step'], d['value'])"
