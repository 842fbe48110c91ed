#!/bin/bash
# training steps, this build against multimodal_amd/lib_base/libmmamd_prev.so (the build of the commit before), alternating; then the gradient tests
export TMPDIR=/tmp; mkdir -p gpurun_out; exec > >(tee gpurun_out/${1:-r06}_train_prev_ab.txt) 2>&1
L=multimodal_amd/lib_base/libmmamd_prev.so
for i in 1 2 3; do
MMAMD_LIB_ALLOW_MISSING=1 MMAMD_LIB=$L python tools/train_bench.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/prev clip-train /'
python tools/train_bench.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new  clip-train /'
done
MMAMD_LIB_ALLOW_MISSING=1 MMAMD_LIB=$L python tools/flava_bench.py --train --steps 6 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/prev flava-train /'
python tools/flava_bench.py --train --steps 6 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new  flava-train /'
MMAMD_LIB_ALLOW_MISSING=1 MMAMD_LIB=$L python tools/coca_bench.py --train 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/prev coca-train /'
python tools/coca_bench.py --train 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new  coca-train /'
python -m pytest tests/test_gpu_backward_kernels.py tests/test_gpu_models.py tests/test_gpu_layer_grad.py tests/test_gpu_coca.py tests/test_gpu_grouped_gemm.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
