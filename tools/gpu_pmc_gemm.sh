#!/bin/bash
# PMC of GEMM variants on the qkv shape: LDS conflicts, waits, MFMA busy (separate passes: <= 8 SQ counters each)
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for v in 0 50; do
  for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $pass | cut -d' ' -f1)
    cd /tmp && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_${v}_$tag -o p -- python $GRAFT_REPO_ROOT/tools/one_gemm.py 50432 2304 768 0 0 $v > /dev/null 2>&1
    f=$(find /tmp/pmc_${v}_$tag -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" $v <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    if "gemm" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    print("variant", sys.argv[2], k, {c: round(v / cnt[(k, c)]) for c, v in agg[k].items()})
PY
  done
done > $O/r2_pmc_gemm_w.txt 2>&1
cat $O/r2_pmc_gemm_w.txt
