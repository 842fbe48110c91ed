#!/bin/bash
# r05: bf16 dgrad hand-over into the LayerNorm backward, again, now that its bf16 instantiation fits three waves per SIMD (alternating processes)
mkdir -p gpurun_out
o=gpurun_out/r05_train_bf16_dh_ab2.txt; : > $o
for r in 1 2 3; do
  python tools/train_bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 >> $o
  python tools/train_bench.py --steps 10 --warmup 3 --f32-dh 2>/dev/null | tail -1 >> $o
done
python -m pytest tests/test_gpu_backward_kernels.py -x -q -m gpu 2>&1 | tail -2 >> $o
cat $o
