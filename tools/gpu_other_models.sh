#!/bin/bash
# r03 re-measurement of the other configurations + FLAVA kernel profile + CLIP training step:  bash tools/gpu_other_models.sh
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3d; mkdir -p $O; cd $GRAFT_REPO_ROOT
{
timeout 300 python tools/clip_fwd_bench.py --model l14 --steps 10 2>/dev/null | tail -1
timeout 300 python tools/clip_fwd_bench.py --model b32 --steps 20 2>/dev/null | tail -1
timeout 300 python tools/clip_fwd_bench.py --model b16 --vision-only --steps 20 2>/dev/null | tail -1
timeout 300 python tools/flava_bench.py 2>/dev/null | tail -1
timeout 300 python tools/coca_bench.py 2>/dev/null | tail -1
timeout 400 python tools/train_bench.py 2>/dev/null | tail -1
} > $O/other_models.jsonl
cat $O/other_models.jsonl | cut -c1-400
cd /tmp && rm -rf /tmp/prof_flava && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_flava -o p -- python $GRAFT_REPO_ROOT/tools/flava_bench.py > $O/flava_rocprof.log 2>&1
f=$(find /tmp/prof_flava -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/flava_kernel_stats.csv && head -14 $O/flava_kernel_stats.csv | cut -c1-200
