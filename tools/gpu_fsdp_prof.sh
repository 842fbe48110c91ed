#!/bin/bash
# r06: where FSDP's extra step time goes: rocprofv3 kernel stats of the FLAVA training step, unwrapped vs wrapped (one RCCL rank)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for arm in plain fsdp; do
  extra=""; [ $arm = fsdp ] && extra="--fsdp"
  cd /tmp && rm -rf /tmp/prof_fl_$arm && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fl_$arm -o fl -- python $R/tools/flava_bench.py --train --steps 4 --warmup 2 $extra > $R/gpurun_out/prof_fl_$arm.log 2>&1
  f=$(find /tmp/prof_fl_$arm -name "*kernel_stats.csv" | head -1)
  t=$(find /tmp/prof_fl_$arm -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $arm "$t" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"== {sys.argv[2]}: kernel time total {tot/1e6:.1f} ms = {tot/1e6/6:.2f} ms per step (6 steps incl. warm-up), {sum(int(r['Calls']) for r in rows)//6} launches per step")
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print(f"    {r['Name'][:84]:84s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:7.2f} ms")
# busy time (union of kernel intervals) of the last 4 steps
tr=list(csv.DictReader(open(sys.argv[3])))
iv=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in tr)
t0=iv[len(iv)//3][0]
iv=[x for x in iv if x[0]>=t0]
busy=0;cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
span=iv[-1][1]-iv[0][0]
print(f"    last 2/3 of the trace: span {span/1e6:.1f} ms, device busy {busy/1e6:.1f} ms ({busy/span*100:.1f} %), idle {100-busy/span*100:.1f} %")
PY
  grep '^{' $R/gpurun_out/prof_fl_$arm.log | grep -o '"ms_per_step": [0-9.]*'
done
