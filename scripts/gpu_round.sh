#!/bin/bash
# One GPU-box visit, staged:  scripts/gpu_round.sh TAG stage [stage ...]
#   tests   pytest -m gpu (full log)               smoke   __graft_entry__.smoke()
#   bench   default bench.py line + per-layer tables (predict + train legs)
#   trace   rocprofv3 --kernel-trace --stats of the same bench command (predict, train, serial train)
#   pmc     counter passes, each its own rocprofv3 run with --kernel-trace only: MFMA/LDS/wait, FETCH_SIZE, WRITE_SIZE
#   sweep   scripts/bench_layer.py tile / row-size sweeps over the benchmark's layers
#   loader  loader-inclusive tiles/s of rs predict / rs train (scripts/loader_bench.py)
#   others  the non-headline configurations, one line each
# Everything lands in gpurun_out/$TAG/ (merged back by gpurun); summaries to keep are copied to profiles/ by hand.
TAG=$1; shift
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou"
# A box whose GPU faults on the first transfer (seen once in round 4: "Memory access fault by GPU node" in every process,
# rocprofv3 then sat in each of its 10-minute timeouts: 55 GPU-minutes gone) must cost seconds, not the visit: one tiny
# device round trip first, and every later command under a timeout sized to its normal duration.
if ! timeout 90 python -c "import torch; x = torch.arange(1 << 20, device='cuda:0', dtype=torch.float32); assert float((x * 2).sum().cpu()) == float((1 << 20) * ((1 << 20) - 1))" > $OUT/sanity.log 2>&1; then
  echo "=== sanity check of the GPU FAILED -- leaving the box"; tail -3 $OUT/sanity.log; exit 3
fi
for STAGE in "$@"; do
  echo "=== stage $STAGE ($(date +%T))"
  case $STAGE in
  tests)
    # -rP: the captured stdout of PASSING tests too (worst / mean gradient cosines, full-size Lovasz errors, mIoU pairs)
    timeout 2400 python -m pytest tests -m gpu -q -rP --maxfail 20 --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest_gpu.log
    grep -E "cosine|Lovasz kernel|mIoU after|main-stream|cfg[0-9] " $OUT/pytest_gpu.log | head -40 ;;
  smoke)
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
  bench)
    timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --layers-json $OUT/layers_predict.json --full-json $OUT/bench_default_full.json > $OUT/bench_default.log 2>&1; echo "bench exit $?"
    tail -1 $OUT/bench_default.log > $OUT/bench_default.json; python scripts/bench_brief.py $OUT/bench_default.json
    timeout 240 $B --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3 --layers-json $OUT/layers_train.json --full-json $OUT/bench_train_full.json > $OUT/bench_train.log 2>&1; echo "train bench exit $?"
    tail -1 $OUT/bench_train.log > $OUT/bench_train_bf16_bs32.json; cut -c1-500 $OUT/bench_train_bf16_bs32.json ;;
  smi)
    # does a concurrent SMU poller (what the driver runs beside its bench: one sample every 5 s) move the train leg?  Same
    # command with a much denser poller (every 0.5 s) beside it; compare train.step_ms min / median / max with `bench`.
    ( while true; do rocm-smi --showuse --showmemuse --json > /dev/null 2>&1; sleep 0.5; done ) & SMI=$!
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-miou --no-cpu-baseline > $OUT/bench_with_smi.log 2>&1; echo "bench exit $?"
    kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
    tail -1 $OUT/bench_with_smi.log > $OUT/bench_with_smi.json; python scripts/bench_brief.py $OUT/bench_with_smi.json ;;
  trace)
    cd /tmp
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_predict -o p -- $B --no-train-leg --steps 5 --warmup 2 > $OUT/trace_predict.log 2>&1; echo "exit $?"
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -o p -- $B --phase train --dtype bf16 --batch 32 --steps 5 --warmup 2 > $OUT/trace_train.log 2>&1; echo "exit $?"
    ROBOSAT_WGRAD_STREAM=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train_serial -o p -- $B --phase train --dtype bf16 --batch 32 --steps 5 --warmup 2 > $OUT/trace_train_serial.log 2>&1; echo "exit $?"
    cd $REPO
    for T in predict train train_serial; do
      F=$(find $OUT/trace_$T -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/${T}_kernel_stats.csv
      find $OUT/trace_$T -name "*kernel_trace.csv" -size +8M -delete
    done ;;
  gaps)
    # where the wall time of one overlapped bf16 train step goes: main-stream busy vs wall, side-stream overlap, idle gaps, copies
    cd /tmp
    timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_gaps -o p -- $B --phase train --dtype bf16 --batch 32 --steps 4 --warmup 2 --no-parity > $OUT/trace_gaps.log 2>&1; echo "exit $?"
    cd $REPO
    F=$(find $OUT/trace_gaps -name "*kernel_trace.csv" | head -1)
    [ -n "$F" ] && python scripts/trace_gaps.py $F stem_fwd > $OUT/train_bf16_bs32_512_trace_gaps.txt 2>&1; head -50 $OUT/train_bf16_bs32_512_trace_gaps.txt | cut -c1-220
    find $OUT/trace_gaps -name "*kernel_trace.csv" -size +8M -delete ;;
  pmc)
    cd /tmp
    C1="SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
    P="$B --no-train-leg --no-parity --steps 1 --warmup 1"
    T="$B --phase train --dtype bf16 --batch 32 --no-parity --steps 1 --warmup 1"
    timeout 200 rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d $OUT/pmc_mfma/predict -o p -- $P > $OUT/pmc_mfma_predict.log 2>&1; echo "exit $?"
    ROBOSAT_WGRAD_STREAM=0 timeout 200 rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d $OUT/pmc_mfma/trainbf16 -o p -- $T > $OUT/pmc_mfma_train.log 2>&1; echo "exit $?"
    for CTR in FETCH_SIZE WRITE_SIZE; do
      timeout 200 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $OUT/pmc/predict_pmc_$CTR -o p -- $P > $OUT/pmc_${CTR}_predict.log 2>&1; echo "exit $?"
      ROBOSAT_WGRAD_STREAM=0 timeout 200 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $OUT/pmc/trainbf16_pmc_$CTR -o p -- $T > $OUT/pmc_${CTR}_train.log 2>&1; echo "exit $?"
    done
    cd $REPO
    find $OUT/pmc $OUT/pmc_mfma -name "*kernel_trace.csv" -size +8M -delete
    python scripts/pmc_mfma_summary.py $OUT/pmc_mfma > $OUT/pmc_mfma_per_kernel.txt 2>&1; head -30 $OUT/pmc_mfma_per_kernel.txt
    python scripts/pmc_traffic.py $OUT/pmc $OUT/pmc_traffic.json "$TAG" > $OUT/pmc_hbm_traffic.txt 2>&1; head -20 $OUT/pmc_hbm_traffic.txt
    du -sh $OUT ;;
  halo)
    bash scripts/halo_sweep.sh > $OUT/halo_sweep.txt 2>&1; cat $OUT/halo_sweep.txt | cut -c1-170 ;;
  newtests)
    # this round's new GPU tests first (a failure here should not cost the whole suite's time)
    timeout 1500 python -m pytest tests/test_gpu_tiles.py -m gpu -q -rP -k "halo" --timeout 600 > $OUT/pytest_halo.log 2>&1; echo "halo tests exit $?"; tail -15 $OUT/pytest_halo.log | cut -c1-300
    timeout 1200 python -m pytest tests/test_gpu_parallel.py -m gpu -q -rP -k "rccl or miou or weighted" --timeout 900 > $OUT/pytest_dp.log 2>&1; echo "dp tests exit $?"; tail -15 $OUT/pytest_dp.log | cut -c1-400 ;;
  rccl1)
    # the RCCL branch of the reducer inside the timed train steps, one GPU (bench.py --force-reducer), fp32 and bf16 wire
    for W in fp32 bf16; do
      timeout 200 $B --phase train --dtype bf16 --batch 32 --steps 10 --warmup 3 --force-reducer --grad-dtype $W --no-parity --full-json $OUT/bench_rccl1_$W.json > $OUT/bench_rccl1_$W.log 2>&1; echo "exit $?"
      tail -1 $OUT/bench_rccl1_$W.log | cut -c1-600
    done ;;
  sweep)
    bash scripts/layer_sweep.sh > $OUT/layer_sweep.txt 2>&1; tail -n 120 $OUT/layer_sweep.txt ;;
  loader)
    timeout 1200 python scripts/loader_bench.py > $OUT/loader_bench.txt 2>&1; echo "exit $?"; cat $OUT/loader_bench.txt | cut -c1-300 ;;
  others)
    {
      for A in "--dtype bf16 --no-train-leg --steps 20" "--phase train --batch 8 --steps 5 --warmup 2" "--size 1024 --batch 8 --no-train-leg --steps 10" "--phase train --dtype bf16 --batch 32 --classes 4 --steps 10" "--size 576 --batch 16 --no-train-leg --steps 10"; do
        timeout 600 $B --no-parity $A 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '->', d['value'], d['unit'], d['ms_per_step'], 'ms/step', d['dtype'], '|', d['config']['workload'])" "$A"
      done
    } > $OUT/bench_others.txt 2>&1
    cat $OUT/bench_others.txt
    timeout 600 python scripts/latency_bench.py > $OUT/latency.txt 2>&1; cat $OUT/latency.txt ;;
  esac
done
echo "=== done ($(date +%T))"
