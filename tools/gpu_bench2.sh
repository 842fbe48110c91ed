#!/bin/bash
set +e
mkdir -p gpurun_out
timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | tee gpurun_out/bench_2stream.log
MMAMD_SINGLE_STREAM=1 timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | tee gpurun_out/bench_1stream.log
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
