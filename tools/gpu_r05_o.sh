#!/bin/bash
# r05: grouped weight gradients (one split-K launch + one reduce per layer) against one launch + reduce per Linear; alternating processes
mkdir -p gpurun_out
o=gpurun_out/r05_train_grouped_wgrad_ab.txt; : > $o
python -m pytest tests/test_gpu_backward_kernels.py tests/test_gpu_layer_grad.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-400 >> $o
for r in 1 2 3; do
  python tools/train_bench.py --steps 10 --warmup 3 --no-grouped-wgrad 2>/dev/null | tail -1 >> $o
  python tools/train_bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 >> $o
done
cat $o | cut -c1-420
