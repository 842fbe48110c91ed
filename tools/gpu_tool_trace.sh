#!/bin/bash
# kernel timeline of any bench tool (all steps): bash tools/gpu_tool_trace.sh <tag> <tool.py> [args]  -> gpurun_out/<tag>_timeline.txt (start us, dur, queue, grid, name)
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; T=$1; shift
cd /tmp && rm -rf /tmp/prof_tt2 && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tt2 -o p -- python $GRAFT_REPO_ROOT/tools/"$@" > $O/${T}_trace.log 2>&1
f=$(find /tmp/prof_tt2 -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/${T}_timeline.txt <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    n=re.sub(r'^void ','',r['Kernel_Name']).replace('mmamd::','')[:80]
    print(f"{(s-t0)/1e3:10.1f} {(e-s)/1e3:7.1f} q{r.get('Queue_Id','?')} {r.get('Grid_Size_X') or r.get('Grid_Size') or '':>8s} {n}")
PY
wc -l $O/${T}_timeline.txt
