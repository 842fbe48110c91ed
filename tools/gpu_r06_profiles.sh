#!/bin/bash
# r06 evidence set (one gpurun call): the whole -m gpu suite, smoke(), the bench lines (default 10 + 50 with the CPU legs, and the driver's 5 + 20), rocprofv3
# kernel stats of the bench command, of the CLIP training step and of the FLAVA forward, the other configurations against the r05 build (same box,
# alternating), PMC passes of the fp32-residual GEMM.  Outputs under gpurun_out/r6p/.
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6p; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -3
cp gpurun_out/parity.json $O/parity.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-300
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.log; cut -c1-400 $O/bench_line.json
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_line_driver_form.json 2>> $O/bench_err.log; cut -c1-200 $O/bench_line_driver_form.json
cd /tmp && rm -rf /tmp/prof_b && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o r6 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-probe > $O/rocprof_bench.log 2>&1
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -8 $O/bench_kernel_stats.csv | cut -c1-160
cd $GRAFT_REPO_ROOT
bash tools/gpu_other_models.sh r6p/other_models > /dev/null 2>&1; mv gpurun_out/r6p/other_models_other_models.jsonl $O/other_models.jsonl 2>/dev/null
grep -o '"arm": "[^"]*", "what": "[^"]*"\|"ms_per_step": [0-9.]*' $O/other_models.jsonl | paste - -
{
timeout 400 python tools/train_bench.py --tower image 2>/dev/null | grep '^{' | tail -1
timeout 400 python tools/train_bench.py --tower text 2>/dev/null | grep '^{' | tail -1
timeout 400 python tools/coca_bench.py --train 2>/dev/null | grep '^{' | tail -1
} > $O/other_models_train_extra.jsonl
cut -c1-200 $O/other_models_train_extra.jsonl
bash tools/gpu_train_stats.sh r6p/r06 > $O/train_stats.txt 2>&1; head -12 $O/train_stats.txt | cut -c1-170
cd /tmp && rm -rf /tmp/prof_flava && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_flava -o p -- python $GRAFT_REPO_ROOT/tools/flava_bench.py > $O/flava_rocprof.log 2>&1
f=$(find /tmp/prof_flava -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/flava_kernel_stats.csv && head -12 $O/flava_kernel_stats.csv | cut -c1-160
cd $GRAFT_REPO_ROOT && bash tools/gpu_pmc_residual.sh > $O/pmc_residual.txt 2>&1; cp gpurun_out/pmc_residual_kernel.json $O/ 2>/dev/null; tail -2 $O/pmc_residual.txt | cut -c1-400
