#!/bin/bash
mkdir -p gpurun_out
L=multimodal_amd/lib_base
python tools/resid_epilogue_ab.py --rounds 3 --extra "stnt=$L/libmmamd_stnt.so,ldnt=$L/libmmamd_ldnt.so,bothnt=$L/libmmamd_bothnt.so,stsc1=$L/libmmamd_stsc1.so,new_stg0=@0,new_stg150=@150,new_stg300=@300,base_stg0=$L/libmmamd_r05.so@0" > gpurun_out/r06_resid_epilogue_ab.txt 2>&1
cat gpurun_out/r06_resid_epilogue_ab.txt
