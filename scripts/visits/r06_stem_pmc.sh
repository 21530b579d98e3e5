#!/bin/bash
# counters of the fp32 predict pass (one step): MFMA busy / LDS / wait / bank conflicts per kernel -- the stem's row is the question
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/r6stempmc; mkdir -p $OUT
B="python $REPO/bench.py --no-cpu-baseline --no-extra-legs --no-miou --no-train-leg --no-parity --steps 1 --warmup 1"
C1="SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d $OUT/predict -o p -- $B > $OUT/pmc.log 2>&1; echo "exit $?"
C2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
timeout 200 rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d $OUT/predict2 -o p -- $B > $OUT/pmc2.log 2>&1; echo "exit $?"
cd $REPO
mkdir -p $OUT/x/predict $OUT/x/trainbf16
cp $(find $OUT/predict -name "p_counter_collection.csv" | head -1) $OUT/x/predict/p_counter_collection.csv
cp $OUT/x/predict/p_counter_collection.csv $OUT/x/trainbf16/p_counter_collection.csv
python scripts/pmc_mfma_summary.py $OUT/x 2>&1 | head -28 | cut -c1-140
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r6stempmc/predict2/**/p_counter_collection.csv', recursive=True)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        if 'stem_conv' in r['Kernel_Name'] or 'conv_wino_f32_kernel' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:60]][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in acc.items():
        print(k, dict(v))
PY
find $OUT -name "*kernel_trace.csv" -size +4M -delete
