#!/bin/bash
# attention backward forms (two kernels | fused two-role | single pass): isolated timings, the parity tests, the training step with each form
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
{
timeout 300 python tools/attn_bwd_bench.py 2>&1 | grep -v amdgpu.ids
echo "# tests"
timeout 400 python -m pytest tests/test_gpu_backward_kernels.py tests/test_gpu_dropout.py tests/test_gpu_coca.py -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
echo "# CLIP training step (tools/train_bench.py), alternating: default | --attn-variant 4000 (two kernels) | 4002 (fused two-role everywhere)"
for i in 1 2; do for v in 4003 4000 4002; do echo -n "variant $v: "; timeout 300 python tools/train_bench.py --attn-variant $v 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done; done
} > $O/r04_attn_bwd_forms.txt 2>&1
cat $O/r04_attn_bwd_forms.txt
