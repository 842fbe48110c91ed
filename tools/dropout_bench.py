"""mmamd_dropout (training-time dropout / stochastic depth pass): isolated timings on the residual-stream shape of a ViT-B/16 B = 256 layer, and a small
encoder-stack training step with and without dropout.      python tools/dropout_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

M, d = 50432, 768
x = torch.randn(M, d, device="cuda")
res = torch.randn(M, d, device="cuda")
xb = x.to(torch.bfloat16)
for name, fn, mb in (("fp32 x + fp32 residual -> fp32 (branch dropout)", lambda: ops.dropout(x, 0.1, 123, 0, residual=res), 3 * M * d * 4),
                     ("fp32 -> bf16 (gradient of a dropped branch)", lambda: ops.dropout(x, 0.1, 123, 0, out_dtype=torch.bfloat16), M * d * 6),
                     ("bf16 in place (MLP hidden, 4 d wide)", None, 0),
                     ("stochastic depth, one decision per sample", lambda: ops.dropout(x, 0.1, 123, 0, residual=res, group=197 * d), 3 * M * d * 4)):
    if fn is None:
        g = torch.randn(M, 4 * d, device="cuda").to(torch.bfloat16)
        fn, mb = (lambda: ops.dropout(g, 0.1, 123, 1, out=g)), M * 4 * d * 4
    us = timeit(fn, 20) * 1e3
    print(f"{name:55s} {us:7.1f} us  {mb / us / 1e6:5.2f} TB/s", flush=True)
