#!/bin/bash
# PMC passes on the dominant GEMM with the CURRENT default variant (separate runs per counter group)
set +e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
cd /tmp
ARGS="50432 3072 768 1 0 0 5"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/one_gemm.py $ARGS > $OUT/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT | tee $OUT/summary.txt
