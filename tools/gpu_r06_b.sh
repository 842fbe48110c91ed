#!/bin/bash
# r06 pass B: fp32-residual epilogue A/B across builds + grouped GEMM tests + bench
mkdir -p gpurun_out
python tools/resid_epilogue_ab.py --rounds 3 > gpurun_out/r06_resid_epilogue_ab.txt 2>&1
cat gpurun_out/r06_resid_epilogue_ab.txt
python -m pytest tests/test_gpu_grouped_gemm.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['loss'], d['roofline']['by_shape'])"
MMAMD_LIB=multimodal_amd/lib_base/libmmamd_r05.so python bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', d['ms_per_step'], d['value'], d['loss'], d['roofline']['by_shape'])"
