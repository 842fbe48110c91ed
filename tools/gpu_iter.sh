#!/bin/bash
# quick iteration pass: kernel tests for gemm + microbench
set +e
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm" --maxfail=10 -p no:cacheprovider 2>&1 | tail -15
timeout 300 python tools/kernel_bench.py --variants ${VARIANTS:-7,18} 2>&1 | tee gpurun_out/kernel_bench.log
