#!/bin/bash
# r05: the pair node (both towers' stacks as one autograd node, grouped launches over both towers) against one node per tower + side stream
mkdir -p gpurun_out
o=gpurun_out/r05_train_pair_node_ab.txt; : > $o
python -m pytest tests/test_gpu_compile_train.py tests/test_gpu_backward_kernels.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | cut -c1-400 >> $o
for r in 1 2 3; do
  python tools/train_bench.py --steps 10 --warmup 3 --no-pair-node 2>/dev/null | tail -1 >> $o
  python tools/train_bench.py --steps 10 --warmup 3 2>&1 | tail -1 >> $o
done
cat $o | cut -c1-330
