#!/bin/bash
# One box visit for the epilogue-wave fp32 1x1 kernel: parity + per-layer timing (scripts/ew_check.py), then the predict pass with it.
OUT=gpurun_out/${1:-ew}; mkdir -p $OUT
timeout 60 python -c "import torch; x = torch.arange(1 << 20, device='cuda:0', dtype=torch.float32); assert float((x * 2).sum().cpu()) == float((1 << 20) * ((1 << 20) - 1))" || { echo "GPU sanity failed"; exit 3; }
timeout 120 python scripts/ew_check.py > $OUT/ew_check.txt 2>&1; echo "ew_check exit $?"; grep -v amdgpu.ids $OUT/ew_check.txt | tail -40
if grep -q "PARITY OK" $OUT/ew_check.txt; then
  B="python bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-train-leg --steps 20 --warmup 5"
  for E in 0 1; do
    if [ $E = 1 ]; then export RS_CONV1X1_EW=1; else unset RS_CONV1X1_EW; fi
    timeout 100 $B --full-json $OUT/full_ew$E.json > $OUT/bench_ew$E.log 2>&1; echo "bench ew=$E exit $?"
    tail -1 $OUT/bench_ew$E.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('EW', sys.argv[1], d['value'], d['unit'], d['ms_per_step'], 'ms', d.get('parity'))" $E
  done
fi
