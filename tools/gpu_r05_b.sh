#!/bin/bash
# r05 call B: the flash + one-pass attention-probability path (csrc/attention_probs_lse.hip): tests, kernel A/B against the two-pass kernel, FLAVA step A/B
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_flava.py tests/test_gpu_attention_ring.py -q -m gpu -x 2>&1 | tail -15 > $O/r05_b_tests.txt
cat $O/r05_b_tests.txt
timeout 300 python tools/probs_lse_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/r05_probs_lse_bench.txt
for i in 1 2; do
  timeout 300 python tools/flava_bench.py --grouped --steps 10 2>/dev/null | tail -1 >> $O/r05_flava_probs_ab.txt
  timeout 300 python tools/flava_bench.py --grouped --steps 10 --probs-two-pass 2>/dev/null | tail -1 >> $O/r05_flava_probs_ab.txt
done
cut -c1-260 $O/r05_flava_probs_ab.txt
