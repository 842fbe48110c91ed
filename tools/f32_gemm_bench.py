"""mmamd_f32_gemm_strided at the shapes of the projection / loss gradients of a CLIP training step.    python tools/f32_gemm_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

for M, N, K, tr in ((768, 512, 256, True), (512, 512, 256, True), (256, 768, 512, False), (256, 512, 512, False), (256, 512, 256, False), (1024, 768, 256, True)):
    X, Y = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    if tr:  # weight gradients: both operands read down their columns (dW = h^T dy)
        Xt, Yt = X.t().contiguous(), Y.t().contiguous()
        us = timeit(lambda: ops.f32_gemm_strided(Xt, 1, M, Yt, 1, N, M, N, K), 50) * 1e3
    else:
        us = timeit(lambda: ops.f32_gemm_strided(X, K, 1, Y, K, 1, M, N, K), 50) * 1e3
    print(f"M={M} N={N} K={K} {'column' if tr else 'row'}-major operands: {us:6.1f} us  ({2.0 * M * N * K / us / 1e6:5.2f} TF/s)", flush=True)
