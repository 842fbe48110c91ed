#!/bin/bash
# r05 call C: second form of the probabilities kernel (no loads in the tile loop, scalar segment arithmetic); FLAVA tests with the bounds restated against the
# reference's own bf16 run; bf16 dh A/B of the CLIP training step with the gradient-fixture test in both modes
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_flava.py tests/test_gpu_headline_parity.py::test_flava_full_size_b16_vs_reference tests/test_gpu_bench_size_parity.py \
  tests/test_gpu_backward_kernels.py::test_clip_training_step_gradients_vs_reference_autograd -q -m gpu -s 2>&1 | grep -v "Warning\|warn" | tail -40 > $O/r05_c_tests.txt
cat $O/r05_c_tests.txt
timeout 300 python tools/probs_lse_bench.py --shapes 256x197x12,128x275x12,256x77x12,256x128x12,256x64x12 2>&1 | grep -v amdgpu.ids | tee $O/r05_probs_lse_bench_v2.txt
for i in 1 2; do
  timeout 300 python tools/flava_bench.py --steps 10 2>/dev/null | tail -1 >> $O/r05_flava_probs_ab.txt
  timeout 300 python tools/flava_bench.py --steps 10 --probs-two-pass 2>/dev/null | tail -1 >> $O/r05_flava_probs_ab.txt
done
cut -c1-300 $O/r05_flava_probs_ab.txt
for i in 1 2; do
done
