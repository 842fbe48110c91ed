#!/bin/bash
# r05 evidence set (one gpurun call): the whole -m gpu suite, smoke(), the bench lines (default 10 + 50, and the driver's 5 + 20), rocprofv3 kernel stats of the
# same command, the other configurations, kernel stats of the CLIP training step and of the FLAVA forward.  Outputs under gpurun_out/r5p/.
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5p; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-300
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench_err.log; cut -c1-400 $O/bench_line.json
timeout 400 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > $O/bench_line_driver_form.json 2>> $O/bench_err.log; cut -c1-200 $O/bench_line_driver_form.json
cd /tmp && rm -rf /tmp/prof_b && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r5 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-probe > $O/rocprof_bench.log 2>&1
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -8 $O/bench_kernel_stats.csv | cut -c1-160
cd $GRAFT_REPO_ROOT
{
timeout 300 python tools/clip_fwd_bench.py --model l14 --steps 10 2>/dev/null | tail -1
timeout 300 python tools/clip_fwd_bench.py --model b32 --steps 20 2>/dev/null | tail -1
timeout 300 python tools/clip_fwd_bench.py --model b16 --vision-only --steps 20 2>/dev/null | tail -1
timeout 300 python tools/flava_bench.py 2>/dev/null | tail -1
timeout 300 python tools/flava_bench.py --no-attentions 2>/dev/null | tail -1
timeout 300 python tools/coca_bench.py 2>/dev/null | tail -1
timeout 400 python tools/train_bench.py 2>/dev/null | tail -1
timeout 400 python tools/train_bench.py --tower image 2>/dev/null | tail -1
timeout 400 python tools/train_bench.py --tower text 2>/dev/null | tail -1
timeout 400 python tools/flava_bench.py --train 2>/dev/null | tail -1
timeout 400 python tools/coca_bench.py --train 2>/dev/null | tail -1
} > $O/other_models.jsonl
cat $O/other_models.jsonl | cut -c1-260
cd /tmp && rm -rf /tmp/prof_t && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o p -- python $GRAFT_REPO_ROOT/tools/train_bench.py > $O/train_rocprof.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/train_kernel_stats.csv && head -14 $O/train_kernel_stats.csv | cut -c1-160
cd /tmp && rm -rf /tmp/prof_flava && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_flava -o p -- python $GRAFT_REPO_ROOT/tools/flava_bench.py > $O/flava_rocprof.log 2>&1
f=$(find /tmp/prof_flava -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/flava_kernel_stats.csv && head -12 $O/flava_kernel_stats.csv | cut -c1-160
