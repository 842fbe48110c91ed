#!/bin/bash
# full GPU pass: all parity tests, microbench, bench line, smoke, rocprof kernel stats (csv) -> gpurun_out/
set +e
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
timeout 300 python tools/kernel_bench.py --variants 0,7 > gpurun_out/kernel_bench.log 2>&1
cat gpurun_out/kernel_bench.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1
cat gpurun_out/bench.log
timeout 300 python tools/flava_bench.py > gpurun_out/flava_bench.log 2>&1
timeout 300 python tools/coca_bench.py > gpurun_out/coca_bench.log 2>&1
cat gpurun_out/coca_bench.log
cat gpurun_out/flava_bench.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
cat gpurun_out/smoke.log
rm -rf gpurun_out/prof
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
grep -h "metric" gpurun_out/rocprof.log | tail -1
find gpurun_out/prof -name "*stats*" | head
timeout 300 python tools/train_bench.py > gpurun_out/train_bench.log 2>&1
tail -1 gpurun_out/train_bench.log
timeout 400 python tools/flava_bench.py --train --steps 5 > gpurun_out/flava_train_bench.log 2>&1
tail -1 gpurun_out/flava_train_bench.log
timeout 400 python tools/coca_bench.py --train --steps 3 --warmup 1 > gpurun_out/coca_train_bench.log 2>&1
tail -1 gpurun_out/coca_train_bench.log
