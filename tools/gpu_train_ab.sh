python -m pytest tests/test_gpu_backward_kernels.py tests/test_gpu_models.py tests/test_gpu_coca.py tests/test_gpu_flava.py tests/test_gpu_layer_grad.py tests/test_gpu_compile_train.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do
MMAMD_LIB_ALLOW_MISSING=1 MMAMD_LIB=multimodal_amd/lib_base/libmmamd_r05.so python tools/train_bench.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/r05 clip-train /'
python tools/train_bench.py 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new clip-train /'
MMAMD_LIB_ALLOW_MISSING=1 MMAMD_LIB=multimodal_amd/lib_base/libmmamd_r05.so python tools/flava_bench.py --train --steps 6 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/r05 flava-train /'
python tools/flava_bench.py --train --steps 6 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new flava-train /'
done
