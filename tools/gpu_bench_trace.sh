#!/bin/bash
# kernel timeline of ONE headline step (the last profiled one): gpurun_out/<tag>_bench_timeline.txt
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && rm -rf /tmp/prof_bt && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_bt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --cpu-sample 0 --no-probe > $O/bench_trace.log 2>&1
f=$(find /tmp/prof_bt -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/${1:-r06}_bench_timeline.txt <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last step: from the last image cast (convert_kernel with the largest grid) to the end
idx=[i for i,r in enumerate(rows) if 'vit_cls_lnpre' in r['Kernel_Name']]
start=idx[-1]; 
while start>0 and 'layernorm_grouped' not in rows[start-1]['Kernel_Name'] and 'gemm_bf16_nt_kernel_ppg' not in rows[start-1]['Kernel_Name']: start-=1
end=len(rows)
last=rows[start:end]
t0=int(last[0]['Start_Timestamp']); prev=t0
def short(s):
    s=re.sub(r'^void ','',s); s=s.replace('mmamd::','')
    return s[:90]
print(f"{len(last)} kernels; span {(int(last[-1]['End_Timestamp'])-t0)/1e6:.3f} ms")
gaps=0
for r in last:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gaps+=max(0,s-prev)
    print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}  gap {(s-prev)/1e3:6.1f}  grid {r.get('Grid_Size_X') or r.get('Grid_Size') or '':>8s}  {short(r['Kernel_Name'])}")
    prev=e
print(f"sum of gaps {gaps/1e3:.1f} us")
PY
head -2 $O/${1:-r06}_bench_timeline.txt; tail -1 $O/${1:-r06}_bench_timeline.txt
