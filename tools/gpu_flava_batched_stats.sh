#!/bin/bash
# FLAVA B=128 with schedule.flava_batched_passes (the default): bench line + rocprofv3 kernel stats.   bash tools/gpu_flava_batched_stats.sh
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3f; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 300 python tools/flava_bench.py 2>/dev/null | tail -1 > $O/flava_batched_line.json; cut -c1-400 $O/flava_batched_line.json
cd /tmp && rm -rf /tmp/prof_flava && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_flava -o p -- python $GRAFT_REPO_ROOT/tools/flava_bench.py > $O/flava_batched_rocprof.log 2>&1
f=$(find /tmp/prof_flava -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/flava_kernel_stats_batched.csv && ls -la $O && head -16 $O/flava_kernel_stats_batched.csv | cut -c1-220
