#!/bin/bash
# one GPU iteration: kernel-level tests + model gradient tests + training bench; everything logged under gpurun_out/
set +e
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_backward_kernels.py -x -q -m gpu > gpurun_out/pytest_iter.log 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_iter.log | tail -3
grep -E "^(FAILED|ERROR)|assert|Error" gpurun_out/pytest_iter.log | head -20
timeout 300 python tools/train_bench.py > gpurun_out/train_bench.log 2>&1
tail -1 gpurun_out/train_bench.log | cut -c1-400
