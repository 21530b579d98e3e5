#!/bin/bash
# Copies what is kept of one GPU-box visit (gpurun_out/TAG, written by scripts/gpu_round.sh) into profiles/ROUND and re-stamps
# profiles/pmc_traffic.json with the commit of the tree that was profiled.   scripts/keep_profiles.sh TAG ROUND [COMMIT]
TAG=$1; ROUND=$2; COMMIT=${3:-$(git rev-parse --short HEAD)}
SRC=gpurun_out/$TAG; DST=profiles/$ROUND; mkdir -p $DST
cp $SRC/bench_default.json $DST/bench_default.json
cp $SRC/bench_default.json $DST/bench_driver_command.json   # (the `bench` stage runs the driver's command: --gpus 1 --steps 20 --warmup 5)
cp $SRC/bench_train_bf16_bs32.json $DST/ 2>/dev/null
# the full records behind the compact stdout lines (per-kernel tables, every step time, counter provenance)
cp $SRC/bench_default_full.json $DST/bench_default_full.json 2>/dev/null
cp $SRC/bench_train_full.json $DST/bench_train_bf16_bs32_full.json 2>/dev/null
# the RCCL branch of the gradient reducer inside the timed steps, one GPU (bench.py --force-reducer)
for W in fp32 bf16; do [ -f $SRC/bench_rccl1_$W.log ] && tail -1 $SRC/bench_rccl1_$W.log > $DST/bench_rccl1_world1_$W.json; done
cp $SRC/layers_predict.json $SRC/layers_train.json $DST/ 2>/dev/null
cp $SRC/predict_kernel_stats.csv $DST/predict_fp32_bs16_512_kernel_stats.csv 2>/dev/null
cp $SRC/train_kernel_stats.csv $DST/train_bf16_bs32_512_kernel_stats.csv 2>/dev/null
cp $SRC/train_serial_kernel_stats.csv $DST/train_bf16_bs32_512_serial_kernel_stats.csv 2>/dev/null
cp $SRC/pmc_mfma_per_kernel.txt $DST/ 2>/dev/null
cp $SRC/bench_others.txt $SRC/latency.txt $DST/ 2>/dev/null
python scripts/bench_brief.py $SRC/bench_default.json > $DST/bench_brief.txt
if [ -d $SRC/pmc ]; then
  # the table stamped ON THE GPU BOX by the pmc stage (digest of the sources that were profiled) is what is kept; this script only adds
  # the commit -- and only if the working tree still has those sources (round 5 re-stamped the digest by hand three times: never again)
  python scripts/pmc_traffic.py $SRC/pmc /tmp/pmc_traffic_here.json $TAG > $DST/pmc_hbm_traffic.txt
  python - "$COMMIT" "$SRC/pmc_traffic.json" <<PY
import json, sys
sys.path.insert(0, ".")
from robosat_amd._lib import kernel_source_digest
d = json.load(open(sys.argv[2]))
m = d["_meta"]
m.setdefault("profiled_csrc_digest", m.get("csrc_digest"))
if m["profiled_csrc_digest"] == kernel_source_digest():
    m["commit"] = sys.argv[1]
else:
    print("WARNING: the working tree's kernel sources (%s) are not the profiled ones (%s): commit left unset, bench.py will withhold roofline.traffic" % (kernel_source_digest(), m["profiled_csrc_digest"]))
json.dump(d, open("profiles/pmc_traffic.json", "w"), indent=1, sort_keys=True)
PY
fi
# the test log without the tool chatter: summary line + every captured line that carries a number a reader may ask for
{ grep -E "passed|failed" $SRC/pytest_gpu.log | tail -1
  grep -E "cosine|Lovasz kernel|lovasz \(|mIoU after|main-stream|cfg[0-9] |max\|dprob\||losses eager|worst " $SRC/pytest_gpu.log | grep -v "print(" | sort -u
} > $DST/pytest_gpu.log
ls -la $DST
