#!/bin/bash
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/train_bench.py > gpurun_out/train_bench.log 2>&1
tail -2 gpurun_out/train_bench.log
rm -rf gpurun_out/prof_train
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -o r01_train -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_train.log 2>&1)
tail -1 gpurun_out/rocprof_train.log
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1)
head -24 $f | cut -c1-150
