#!/bin/bash
mkdir -p gpurun_out
L=multimodal_amd/lib_base
python tools/resid_epilogue_ab.py --rounds 3 --extra "w1sc1=$L/libmmamd_w1sc1.so,w2=$L/libmmamd_w2.so" > gpurun_out/r06_resid_epilogue_ab2.txt 2>&1
cat gpurun_out/r06_resid_epilogue_ab2.txt
bash tools/bench_libs_ab.sh 3 base=$L/libmmamd_r05.so new= w1sc1=$L/libmmamd_w1sc1.so 2>&1 | tee gpurun_out/r06_epilogue_step_ab2.txt
