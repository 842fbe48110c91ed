#!/bin/bash
# first GPU pass: parity tests, kernel microbench, bench line, rocprof kernel stats
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 300 python tools/kernel_bench.py > gpurun_out/kernel_bench.log 2>&1
cat gpurun_out/kernel_bench.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1
cat gpurun_out/bench.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
cat gpurun_out/smoke.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -5 gpurun_out/rocprof.log
find gpurun_out/prof -name "*stats*" | head
