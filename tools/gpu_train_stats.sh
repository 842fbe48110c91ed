#!/bin/bash
# rocprofv3 kernel stats of the CLIP training step (7 profiled steps)
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && rm -rf /tmp/prof_t && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o p -- python $GRAFT_REPO_ROOT/tools/train_bench.py > $O/train_rocprof.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${1:-r06}_train_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over 7 steps = {tot/7e6:.2f} ms per step")
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:26]:
    print(f"{r['Name'][:104]:104s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/tot*100:5.1f} %")
PY
