for V in "RS_CONV_BIG=3" "RS_CONV_BIG=0" "RS_CONV_BIG=1" "RS_CONV_BIG=2" "RS_CONV_BIG=3" "RS_CONV_BIG=0"; do echo "== [$V]"
env $V timeout 300 python bench.py --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
