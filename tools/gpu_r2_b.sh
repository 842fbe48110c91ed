#!/bin/bash
# round 2, GPU call B: LN-fold kernels + stack parity, model parity with the fold on, bench fold on / off, kernel stats
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_lnfold.py tests/test_gpu_round2_boundary.py -q -m gpu -x > $O/r2b_pytest_lnfold.log 2>&1; tail -4 $O/r2b_pytest_lnfold.log
timeout 1500 python -m pytest tests/test_gpu_headline_parity.py tests/test_gpu_models.py -q -m gpu > $O/r2b_pytest_models.log 2>&1; tail -4 $O/r2b_pytest_models.log
cp $O/r02_parity.json $O/r02_parity_lnfold.json 2>/dev/null
timeout 300 python bench.py --steps 50 --warmup 10 --cpu-sample 0 > $O/r2b_bench_fold.log 2>&1; echo "fold: $(tail -1 $O/r2b_bench_fold.log | cut -c1-260)"
MMAMD_LN_FOLD=0 timeout 300 python bench.py --steps 50 --warmup 10 --cpu-sample 0 > $O/r2b_bench_nofold.log 2>&1; echo "nofold: $(tail -1 $O/r2b_bench_nofold.log | cut -c1-260)"
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o r2b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-probe > $GRAFT_REPO_ROOT/$O/r2b_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r2b_kernel_stats.csv && head -30 $O/r2b_kernel_stats.csv | cut -c1-200
