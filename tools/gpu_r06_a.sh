#!/bin/bash
# r06 pass A: the new W = 8 loss tests, the -inf cross-entropy test, and a bench line of this box
mkdir -p gpurun_out
python -m pytest tests/test_gpu_loss_w8.py tests/test_gpu_backward_kernels.py::test_cross_entropy_with_masked_minus_inf_logits tests/test_gpu_backward_kernels.py::test_cross_entropy_backward_kernel -x -q -m gpu > gpurun_out/r06_a_tests.txt 2>&1
tail -15 gpurun_out/r06_a_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_a_bench.json 2> gpurun_out/r06_a_bench.err
cat gpurun_out/r06_a_bench.json | head -c 600
