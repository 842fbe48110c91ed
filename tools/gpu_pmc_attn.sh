#!/bin/bash
# PMC passes on the attention forward kernel (vision shape); each counter group in its own run, --pmc never combined with traces
set +e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/one_attn.py > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT attention_fwd_kernel | tee $OUT/summary.txt
