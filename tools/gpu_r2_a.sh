#!/bin/bash
# round 2, GPU call A: placement census, new parity / boundary tests, bench baseline, CU-partition sweep, --gpus 2 failure mode
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out
timeout 300 python tools/cu_census.py > $O/r2a_census.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_headline_parity.py tests/test_gpu_round2_boundary.py tests/test_gpu_loss_grad.py -q -m gpu > $O/r2a_pytest.log 2>&1
tail -5 $O/r2a_pytest.log
timeout 600 python bench.py --steps 50 --warmup 10 > $O/r2a_bench.log 2>&1
tail -1 $O/r2a_bench.log | cut -c1-400
for t in 3 4 6; do
  MMAMD_CU_SPLIT=$t timeout 300 python bench.py --steps 30 --warmup 5 --cpu-sample 0 > $O/r2a_bench_split${t}.log 2>&1
  echo "split $t: $(tail -1 $O/r2a_bench_split${t}.log | cut -c1-200)"
done
MMAMD_CU_SPLIT=4 MMAMD_CU_LAYOUT=contiguous timeout 300 python bench.py --steps 30 --warmup 5 --cpu-sample 0 > $O/r2a_bench_split4c.log 2>&1
echo "split 4 contiguous: $(tail -1 $O/r2a_bench_split4c.log | cut -c1-200)"
MMAMD_SINGLE_STREAM=1 timeout 300 python bench.py --steps 30 --warmup 5 --cpu-sample 0 > $O/r2a_bench_1stream.log 2>&1
echo "single stream: $(tail -1 $O/r2a_bench_1stream.log | cut -c1-200)"
timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/r2a_bench_gpus2.log 2>&1; echo "gpus2 rc=$?"; tail -3 $O/r2a_bench_gpus2.log
cat $O/r2a_census.log
