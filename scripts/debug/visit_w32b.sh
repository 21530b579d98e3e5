B="python bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-parity"
for NB in 1024 512 2048 768 1024; do
  RS_WGRAD_F32_BLOCKS=$NB timeout 600 $B --phase train --dtype fp32 --batch 8 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('blocks=$NB fp32 train bs8', d['value'], d['ms_per_step'], d['step_ms']['median'])"
done
