#!/bin/bash
# GPU pass for the input side: parity tests of mmamd_image_resample + the loader micro-benchmark + its kernel stats
set +e
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_transforms.py -q -m gpu > gpurun_out/pytest_transforms.log 2>&1
grep -E "passed|failed|error" gpurun_out/pytest_transforms.log | tail -3
grep -E "^(FAILED|ERROR)|assert|Error" gpurun_out/pytest_transforms.log | head -30
timeout 300 python tools/transform_bench.py > gpurun_out/transform_bench.log 2>&1
tail -1 gpurun_out/transform_bench.log | cut -c1-900
bash tools/gpu_transforms_prof.sh
