#!/bin/bash
# How sensitive is the bf16 train step (BASELINE configs[2]) to a slow host?  The step is ~600 launches issued by Python in
# ~12 ms against ~23 ms of GPU work: a host 2x slower makes it host-bound.  Runs the train bench (a) as is, (b) with the
# process pinned to ONE core that a busy loop shares (the host thread gets ~half a core), each eager and as a hipGraph replay,
# (c) the graph captured without the weight-gradient side stream.   bash scripts/host_sensitivity.sh OUTDIR
OUT=$1; mkdir -p $OUT
B="python bench.py --phase train --dtype bf16 --batch 32 --steps 30 --warmup 5 --no-parity --no-cpu-baseline"
brief() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['ms_per_step'], 'ms/step', d['step_ms'], 'hipgraph', d.get('hipgraph'))" "$2"; }
ROBOSAT_TRAIN_GRAPH=0 timeout 600 $B > $OUT/hs_eager.log 2>&1; brief $OUT/hs_eager.log "eager, free host:        "
ROBOSAT_TRAIN_GRAPH=1 timeout 600 $B > $OUT/hs_graph.log 2>&1; brief $OUT/hs_graph.log "hipGraph, free host:     "
ROBOSAT_TRAIN_GRAPH=1 ROBOSAT_WGRAD_STREAM=0 timeout 600 $B > $OUT/hs_graph_serial.log 2>&1; brief $OUT/hs_graph_serial.log "hipGraph 1 stream, free: "
ROBOSAT_TRAIN_GRAPH=0 ROBOSAT_WGRAD_STREAM=0 timeout 600 $B > $OUT/hs_eager_serial.log 2>&1; brief $OUT/hs_eager_serial.log "eager 1 stream, free:    "
( taskset -c 3 sh -c 'while :; do :; done' ) & SPIN=$!
ROBOSAT_TRAIN_GRAPH=0 timeout 900 taskset -c 3 $B > $OUT/hs_eager_loaded.log 2>&1; brief $OUT/hs_eager_loaded.log "eager, shared core:      "
ROBOSAT_TRAIN_GRAPH=1 timeout 900 taskset -c 3 $B > $OUT/hs_graph_loaded.log 2>&1; brief $OUT/hs_graph_loaded.log "hipGraph, shared core:   "
kill $SPIN 2>/dev/null; wait $SPIN 2>/dev/null
timeout 300 python scripts/host_overhead.py --fused > $OUT/host_overhead.txt 2>&1; tail -4 $OUT/host_overhead.txt
