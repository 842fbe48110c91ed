#!/bin/bash
# PMC of the attention kernels (vision shape): $1 = list of mmamd_debug_set_attn_variant values (0 ring, 1000 r02, 2000+abl ablations)
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3a; mkdir -p $O
for v in ${1:-0 1000}; do
  for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES"; do
    tag=$(echo $pass | cut -d' ' -f1)
    cd /tmp && timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmca_${v}_$tag -o p -- python $GRAFT_REPO_ROOT/tools/one_attn.py $v > /dev/null 2>&1
    f=$(find /tmp/pmca_${v}_$tag -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" $v <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:50]
    if "attention" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    print("variant", sys.argv[2], k, {c: round(v / cnt[(k, c)]) for c, v in agg[k].items()})
PY
  done
done
