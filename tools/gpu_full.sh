#!/bin/bash
# full GPU test suite + smoke + bench (+ kernel stats)
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu > $O/r2_pytest_gpu.log 2>&1; grep -E "passed|failed" $O/r2_pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r2_smoke.log 2>&1; tail -2 $O/r2_smoke.log
timeout 600 python bench.py > $O/r2_bench.log 2>&1; tail -1 $O/r2_bench.log | cut -c1-420
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o r2 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-probe > $O/r2_rocprof.log 2>&1
f=$(find /tmp/prof_full -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r2_kernel_stats.csv && head -24 $O/r2_kernel_stats.csv | cut -c1-170
