#!/bin/bash
# Does the HIP queue priority of the two streams of the train step (main: forward / dgrad / BatchNorm chain, side: weight
# (ROBOSAT_MAIN_PRIORITY / ROBOSAT_SIDE_PRIORITY were two-line measurement knobs in bench.py / autograd._side_stream -- torch.cuda.Stream(priority=...) --
#  not kept: the result, "no effect", is in profiles/r04/halo_variants.txt.)
# gradients) move the bf16 train step?  ROBOSAT_MAIN_PRIORITY (bench.py) / ROBOSAT_SIDE_PRIORITY (autograd._side_stream).
OUT=gpurun_out/${1:-prio}; mkdir -p $OUT
timeout 60 python -c "import torch; x = torch.arange(1 << 20, device='cuda:0', dtype=torch.float32); assert float((x * 2).sum().cpu()) == float((1 << 20) * ((1 << 20) - 1)); print(torch.cuda.Stream.priority_range())" || { echo "GPU sanity failed"; exit 3; }
B="python bench.py --no-cpu-baseline --no-extra-legs --no-miou --phase train --dtype bf16 --batch 32 --steps 20 --warmup 5 --no-parity"
run() {
  local T=$1; shift
  env "$@" timeout 200 $B --full-json $OUT/full_$T.json > $OUT/bench_$T.log 2> $OUT/err_$T.log; echo "bench $T exit $?"; grep "priority" $OUT/err_$T.log
  tail -1 $OUT/bench_$T.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['value'], d['unit'], d['ms_per_step'], 'ms', d.get('step_ms'))" $T
}
run base A=0
run main_hi ROBOSAT_MAIN_PRIORITY=-1
run side_lo ROBOSAT_SIDE_PRIORITY=1
run side_hi ROBOSAT_SIDE_PRIORITY=-1
run base2 A=0
