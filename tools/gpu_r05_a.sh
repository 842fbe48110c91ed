#!/bin/bash
# r05 call A: new tests (post-norm / masks / stand-alone layers / hooked stacks, FSDP, fused bias gradient, eval-mode dropout), training-step A/B of the
# fused bias gradient (alternating processes), MFMA power microbenchmark under telemetry
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_layer_grad.py tests/test_gpu_fsdp_single_rank.py "tests/test_gpu_backward_kernels.py::test_weight_gradient_gemm_with_fused_bias_gradient" "tests/test_gpu_dropout.py::test_eval_mode_stack_with_grad_input_applies_no_dropout" -q -m gpu 2>&1 | tail -40 > $O/r05_a_tests.txt
cat $O/r05_a_tests.txt | tail -25
for i in 1 2; do
  python tools/train_bench.py --steps 8 --warmup 3 2>/dev/null | tail -1 >> $O/r05_train_fused_bias_ab.txt
  python tools/train_bench.py --steps 8 --warmup 3 --no-fused-bias 2>/dev/null | tail -1 >> $O/r05_train_fused_bias_ab.txt
done
cat $O/r05_train_fused_bias_ab.txt | cut -c1-330
hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_power tools/microbench/mfma_power.hip 2>/dev/null
python tools/power_clocks.py --hz 20 --out $O/r05_mfma_power.json --cmd "mfma_zeros=/tmp/mfma_power 0 5" --cmd "mfma_const=/tmp/mfma_power 1 5" --cmd "mfma_random_same_operands=/tmp/mfma_power 2 5" --cmd "mfma_random_gemm_like=/tmp/mfma_power 3 5" --cmd "mfma_random_gemm_like_1wave=/tmp/mfma_power 3 5 1" 2>&1 | grep -v amdgpu.ids | tee $O/r05_mfma_power.txt
