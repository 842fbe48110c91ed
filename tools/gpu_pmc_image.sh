#!/bin/bash
# PMC passes (L2-miss fetch / write bytes, LDS activity) on the image resample kernels; one counter group per run
set +e
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_image
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/transform_bench.py --iters 3 --cpu-sample 1 > $OUT/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
