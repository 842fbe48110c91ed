"""How the persistent GEMM scales with the number of CUs it may use (mmamd_stream_set_cus: the grid of a persistent launch): the four projection shapes of the
ViT-B/16 B = 256 layer on 64 / 128 / 192 / 256 CUs, nothing else running.  Perfect scaling = time x CUs constant; what is lost towards 256 CUs is what the CUs
cost each other (memory system, clock).      python tools/gemm_cu_budget_sweep.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

M = 50432
st = torch.cuda.current_stream()
for name, N, K, act, res in (("qkv", 2304, 768, ops.ACT_NONE, False), ("out+res", 768, 768, ops.ACT_NONE, True), ("up+gelu", 3072, 768, ops.ACT_QUICKGELU, False),
                             ("down+res", 768, 3072, ops.ACT_NONE, True)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    out = torch.randn(M, N, device="cuda") if res else torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.set_gemm_variant(18)  # the persistent kernel, no row-range split
    line = f"{name:9s}"
    base = None
    for cus in (256, 192, 128, 64, 256):
        ops.stream_set_cus(st, 0 if cus == 256 else cus)
        us = timeit(lambda: ops.gemm_bf16(a, w, b, act=act, residual=out if res else None, out=out), 10) * 1e3
        base = base or us
        line += f" | {cus:3d} CUs {us:7.1f} us  x CUs/256 = {us * cus / 256:6.1f}"
    ops.stream_set_cus(st, 0)
    ops.set_gemm_variant(0)
    print(line, flush=True)
