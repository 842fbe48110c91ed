#!/bin/bash
# kernel timeline of ONE CLIP training step (the last profiled one): name, grid, duration, gap to the previous kernel -> gpurun_out/<tag>_train_timeline.txt
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && rm -rf /tmp/prof_tt && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tt -o p -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 3 --warmup 2 > $O/train_trace.log 2>&1
f=$(find /tmp/prof_tt -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/${1:-r06}_train_timeline.txt <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows)//5
last=rows[-n:]
t0=int(last[0]['Start_Timestamp']); prev=t0
def short(s):
    s=re.sub(r'^void ','',s); s=s.replace('mmamd::','')
    return s[:86]
print(f"{n} kernels per step; step span {(int(last[-1]['End_Timestamp'])-t0)/1e6:.2f} ms")
# per-queue occupancy over the window: busy time (union of its kernels) and the idle gaps between consecutive kernels of the SAME queue
qs={}
for r in last: qs.setdefault(r.get('Queue_Id','?'),[]).append((int(r['Start_Timestamp']),int(r['End_Timestamp'])))
for q,iv in qs.items():
    iv.sort(); busy=sum(e-s for s,e in iv); gaps=[iv[i+1][0]-iv[i][1] for i in range(len(iv)-1)]
    big=sorted([g for g in gaps if g>0],reverse=True)[:5]
    print(f"queue {q}: {len(iv)} kernels, busy {busy/1e6:.2f} ms, span {(iv[-1][1]-iv[0][0])/1e6:.2f} ms, idle inside {sum(g for g in gaps if g>0)/1e6:.2f} ms, largest gaps us {[round(g/1e3,1) for g in big]}")
for r in last:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    g=r.get('Grid_Size_X') or r.get('Grid_Size') or ''
    print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  gap {(s-prev)/1e3:6.1f}  q {r.get('Queue_Id','?'):>2s}  grid {g:>8s}  {short(r['Kernel_Name'])}")
    prev=e
PY
head -5 $O/${1:-r06}_train_timeline.txt
