#!/bin/bash
mkdir -p gpurun_out
L=multimodal_amd/lib_base
python tools/resid_epilogue_ab.py --rounds 3 > gpurun_out/r06_resid_epilogue_ab3.txt 2>&1
cat gpurun_out/r06_resid_epilogue_ab3.txt
bash tools/bench_libs_ab.sh 3 base=$L/libmmamd_r05.so new= 2>&1 | tee gpurun_out/r06_epilogue_step_ab3.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r06_e_tests.txt
