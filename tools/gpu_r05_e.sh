#!/bin/bash
# r05 call E: third form of the probabilities kernel (linear band image, ds_read_b128 -> global_store_dwordx4): tests, ablations, kernel and FLAVA step A/B
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_flava.py -q -m gpu 2>&1 | grep -v "Warning\|warn" | tail -15 > $O/r05_e_tests.txt
cat $O/r05_e_tests.txt
timeout 300 python tools/probs_lse_ablate.py 2>&1 | grep -v amdgpu.ids | tee $O/r05_probs_lse_ablation_v3.txt
timeout 300 python tools/probs_lse_bench.py --shapes 256x197x12,128x275x12,256x77x12,256x129x12,256x65x12 2>&1 | grep -v amdgpu.ids | tee $O/r05_probs_lse_bench_v3.txt
rm -f $O/r05_flava_probs_ab.txt
for i in 1 2; do
  timeout 300 python tools/flava_bench.py --steps 10 2>/dev/null | tail -1 >> $O/r05_flava_probs_ab.txt
  timeout 300 python tools/flava_bench.py --steps 10 --probs-two-pass 2>/dev/null | tail -1 >> $O/r05_flava_probs_ab.txt
done
cut -c1-300 $O/r05_flava_probs_ab.txt
