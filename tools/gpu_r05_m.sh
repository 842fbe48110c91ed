#!/bin/bash
# r05 call M: deferred LayerNorm-backward reductions (one batched launch per stack): tests + same-box A/B of the three training steps
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_backward_kernels.py tests/test_gpu_layer_grad.py tests/test_gpu_dropout.py tests/test_gpu_compile_train.py -q -m gpu 2>&1 | tail -5 > $O/r05_m_tests.txt
cat $O/r05_m_tests.txt
rm -f $O/r05_train_deferred_ln_reduce_ab.txt
for i in 1 2; do
  timeout 300 python tools/train_bench.py --steps 8 --warmup 3 2>/dev/null | tail -1 >> $O/r05_train_deferred_ln_reduce_ab.txt
  timeout 300 python tools/train_bench.py --steps 8 --warmup 3 --no-deferred-ln-reduce 2>/dev/null | tail -1 >> $O/r05_train_deferred_ln_reduce_ab.txt
done
cut -c90-420 $O/r05_train_deferred_ln_reduce_ab.txt
