#!/bin/bash
# rocprofv3 kernel stats of the loader micro-benchmark (input side)
set +e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tr -o tr -- python $GRAFT_REPO_ROOT/tools/transform_bench.py --iters 10 > $GRAFT_REPO_ROOT/gpurun_out/transform_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_tr -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/transform_kernel_stats.csv
head -8 gpurun_out/transform_kernel_stats.csv | cut -c1-220
find gpurun_out/prof_tr -name "*kernel_trace.csv" -delete
tail -1 gpurun_out/transform_prof.log | cut -c1-600
