export TMPDIR=/tmp
REPO=$(pwd); cd /tmp
for V in "RS_CONV_TILE=6 RS_CONV_ROWB=128" "RS_CONV_TILE=0 RS_CONV_ROWB=128"; do
T=$(echo $V | tr ' =' '__')
env $V timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $REPO/gpurun_out/pmc_mfma_$T -o p -- python $REPO/scripts/bench_layer.py --bf16 --iters 5 32,1280,64,64,256,3,1,1 > $REPO/gpurun_out/pmc_mfma_$T.log 2>&1; echo "exit $?"; tail -2 $REPO/gpurun_out/pmc_mfma_$T.log
done
