#!/bin/bash
# r05: timeline of the CLIP training step (kernel trace -> busy / idle per queue)
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && rm -rf /tmp/tl && timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 6 --warmup 3 > $O/tl_run.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1); ls -la $f; head -2 $f | cut -c1-400
cd $GRAFT_REPO_ROOT && python tools/step_timeline.py $f --last-ms 250 > $O/r05_train_step_timeline.txt 2>&1; cat $O/r05_train_step_timeline.txt | cut -c1-220
tail -1 $O/tl_run.log | cut -c1-200
