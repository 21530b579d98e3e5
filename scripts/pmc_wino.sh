#!/bin/bash
# SQ counters of the Winograd / phase DecoderBlock launches (scripts/bench_wino.py): usage scripts/pmc_wino.sh OUTDIR
OUT=$1; shift
export TMPDIR=/tmp
REPO=$(pwd); mkdir -p $OUT; cd /tmp
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $REPO/$OUT/sq1 -o p -- python $REPO/scripts/bench_wino.py --iters 2 > $REPO/$OUT/sq1.log 2>&1
C2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_WAVES SQ_INST_CYCLES_VMEM"
timeout 300 rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d $REPO/$OUT/sq2 -o p -- python $REPO/scripts/bench_wino.py --iters 2 > $REPO/$OUT/sq2.log 2>&1
cd $REPO
python - <<PY
import csv,collections,glob
for d in ("sq1","sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:64]+" grid="+r.get("Grid_Size","?")
            if "conv_" not in k: continue
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]);
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES","SQ_INSTS_VALU"): n[k]+=1
        for k,v in acc.items():
            print(k, "launches", n[k])
            for c,val in sorted(v.items()): print("    %-28s %14.0f  per launch %12.0f"%(c,val,val/max(n[k],1)))
PY
