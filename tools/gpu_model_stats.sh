#!/bin/bash
# rocprofv3 kernel stats of any bench tool:  bash tools/gpu_model_stats.sh <tag> <tool.py> [args...]   -> gpurun_out/<tag>_kernel_stats.csv + top kernels
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; T=$1; shift
cd /tmp && rm -rf /tmp/prof_m && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o p -- python $GRAFT_REPO_ROOT/tools/"$@" > $O/${T}_rocprof.log 2>&1
f=$(find /tmp/prof_m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${T}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:22]:
    print(f"{r['Name'][:110]:110s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/tot*100:5.1f} %")
PY
