#!/bin/bash
# r06: attention backward kernels, this build against the r05 build (isolated timings), the gradient tests, and the training steps
export TMPDIR=/tmp; mkdir -p gpurun_out; exec > >(tee gpurun_out/r06_attn_bwd_ab.txt) 2>&1
L=multimodal_amd/lib_base/libmmamd_r05.so
for i in 1 2; do
echo "== r05 build"; MMAMD_LIB_ALLOW_MISSING=1 MMAMD_LIB=$L python tools/attn_bwd_bench.py 2>/dev/null | grep " us"
echo "== this build"; python tools/attn_bwd_bench.py 2>/dev/null | grep " us"
done
python -m pytest tests/test_gpu_backward_kernels.py tests/test_gpu_models.py tests/test_gpu_layer_grad.py tests/test_gpu_coca.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2; do
MMAMD_LIB_ALLOW_MISSING=1 MMAMD_LIB=$L python tools/train_bench.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/r05 clip-train /'
python tools/train_bench.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new clip-train /'
done
MMAMD_LIB_ALLOW_MISSING=1 MMAMD_LIB=$L python tools/flava_bench.py --train --steps 6 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/r05 flava-train /'
python tools/flava_bench.py --train --steps 6 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new flava-train /'
MMAMD_LIB_ALLOW_MISSING=1 MMAMD_LIB=$L python tools/coca_bench.py --train 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/r05 coca-train /'
python tools/coca_bench.py --train 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new coca-train /'
