#!/bin/bash
# FLAVA / CoCa timing + rocprof kernel stats -> gpurun_out/
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/coca_bench.py > gpurun_out/coca_bench.log 2>&1
cat gpurun_out/coca_bench.log | tail -3
for w in flava coca; do
  rm -rf gpurun_out/prof_$w
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$w -o r01_$w -- python $GRAFT_REPO_ROOT/tools/${w}_bench.py --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$w.log 2>&1)
  tail -1 gpurun_out/rocprof_$w.log
  f=$(find gpurun_out/prof_$w -name "*kernel_stats.csv" | head -1)
  head -14 $f | cut -c1-170
done
