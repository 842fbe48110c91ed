#!/bin/bash
set +e
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_zero_shot.py tests/test_gpu_transforms.py -q -m gpu > gpurun_out/pytest_zs.log 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_zs.log | tail -3
grep -E "^(FAILED|ERROR)|assert|Error|^E " gpurun_out/pytest_zs.log | head -40
