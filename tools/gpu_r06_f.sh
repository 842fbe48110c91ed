#!/bin/bash
mkdir -p gpurun_out
L=multimodal_amd/lib_base
python tools/resid_epilogue_ab.py --rounds 3 > gpurun_out/r06_drain_ab.txt 2>&1
cat gpurun_out/r06_drain_ab.txt
bash tools/bench_libs_ab.sh 3 base=$L/libmmamd_r05.so new= 2>&1 | tee gpurun_out/r06_drain_step_ab.txt
python -m pytest tests/test_gpu_grouped_gemm.py tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_headline_parity.py tests/test_gpu_bench_size_parity.py -x -q -m gpu 2>&1 | tail -5
