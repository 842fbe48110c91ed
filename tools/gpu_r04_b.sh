#!/bin/bash
# r04 call B: full GPU suite, per-kernel stats of the base and the phased schedule, the new bench line
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > $O/r04b_gpu_suite.txt 2>&1; grep -n "passed\|failed" $O/r04b_gpu_suite.txt | tail -3
bash tools/gpu_step_stats.sh "base phases2+lead2 phases2+lead6" r04b > $O/r04b_step_stats.txt 2>&1; cat $O/r04b_step_stats.txt | cut -c1-150
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > $O/r04b_bench.json 2> $O/r04b_bench.err; cat $O/r04b_bench.json | cut -c1-1500
