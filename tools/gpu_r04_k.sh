#!/bin/bash
# r04 call K: training step A/B (attention backward form, tile order), FLAVA grouped A/B, decoder dropout test
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dropout.py -q -k "decoder" 2>&1 | tail -3
for i in 1 2; do
  python tools/train_bench.py --steps 8 2>/dev/null | tail -1 | cut -c1-200
  python tools/train_bench.py --steps 8 --attn-variant 4000 2>/dev/null | tail -1 | cut -c1-200
  python tools/train_bench.py --steps 8 --gemm-gm 8 2>/dev/null | tail -1 | cut -c1-200
done > $O/r04k_train_ab.txt 2>&1; cat $O/r04k_train_ab.txt
for i in 1 2; do
  python tools/flava_bench.py --steps 10 2>/dev/null | tail -1 | cut -c1-220
  python tools/flava_bench.py --steps 10 --grouped 2>/dev/null | tail -1 | cut -c1-220
done > $O/r04k_flava_ab.txt 2>&1; cat $O/r04k_flava_ab.txt
python -m pytest tests/test_gpu_flava.py tests/test_gpu_bench_size_parity.py -q 2>&1 | tail -3
