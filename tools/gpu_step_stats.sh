#!/bin/bash
# rocprofv3 kernel stats of the headline step under the given arms of tools/step_ab.py:  bash tools/gpu_step_stats.sh "base delta_ln" [tag]
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${2:-r3b}; mkdir -p $O
for arm in ${1:-base}; do
  cd /tmp && rm -rf /tmp/prof_$arm && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$arm -o p -- python $GRAFT_REPO_ROOT/tools/step_ab.py --arms $arm --rounds 1 --steps 25 --no-graph > $O/stats_$arm.log 2>&1
  f=$(find /tmp/prof_$arm -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/kernel_stats_$arm.csv && python3 - "$f" $arm <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"== {sys.argv[2]}: total kernel time {tot/1e6:.1f} ms over the run (29 steps incl. warm-up + result pass)")
for r in rows[:16]:
    print(f"{r['Name'][:86]:86s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
PY
done
