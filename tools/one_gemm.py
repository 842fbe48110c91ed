#!/usr/bin/env python
"""Run ONE GEMM launch a few times (for rocprofv3 --pmc passes):  python tools/one_gemm.py M N K act res variant [iters [M2 N2 K2]]
(M2 N2 K2: a second problem of the same epilogue kind -> one GROUPED launch, ops.gemm_bf16_grouped)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402

M, N, K, act, res, variant = (int(x) for x in sys.argv[1:7])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
w = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16)
bias = torch.randn(N, generator=g).to(dev)
out = torch.zeros((M, N), dtype=torch.float32 if res else torch.bfloat16, device=dev)
ops.set_gemm_variant(variant)
import os  # noqa: E402
if os.environ.get("MMAMD_GEMM_CN"):  # column tiles per chunk of the tile order (-1 = no chunking): PMC A/B of tools/gpu_pmc_cn.sh
    from multimodal_amd import _lib

    _lib.lib().mmamd_debug_set_gemm_knob(4, int(os.environ["MMAMD_GEMM_CN"]))
if len(sys.argv) > 10:
    M2, N2, K2 = (int(x) for x in sys.argv[8:11])
    a2 = torch.randn(M2, K2, generator=g).to(dev).to(torch.bfloat16)
    w2 = (torch.randn(N2, K2, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    bias2 = torch.randn(N2, generator=g).to(dev)
    out2 = torch.zeros((M2, N2), dtype=out.dtype, device=dev)
    for _ in range(iters):
        ops.gemm_bf16_grouped([(a, w, bias, out if res else None, out), (a2, w2, bias2, out2 if res else None, out2)], act=act, out_dtype=out.dtype)
else:
    for _ in range(iters):
        ops.gemm_bf16(a, w, bias, act=act, residual=out if res else None, out=out)
torch.cuda.synchronize()
