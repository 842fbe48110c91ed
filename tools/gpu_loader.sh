#!/bin/bash
# input side: parity tests, loader micro-benchmark, raw-input end-to-end bench
set +e
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_transforms.py tests/test_gpu_zero_shot.py -q -m gpu > gpurun_out/pytest_transforms.log 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_transforms.log | tail -3
grep -E "^(FAILED|ERROR)|assert|Error" gpurun_out/pytest_transforms.log | head -20
timeout 300 python tools/transform_bench.py > gpurun_out/transform_bench.log 2>&1
tail -1 gpurun_out/transform_bench.log | cut -c1-900
timeout 300 python tools/e2e_bench.py > gpurun_out/e2e_bench.log 2>&1
tail -1 gpurun_out/e2e_bench.log | cut -c1-900
