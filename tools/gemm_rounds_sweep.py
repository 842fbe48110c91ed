#!/usr/bin/env python
"""Is the partial last round of a persistent GEMM launch idle capacity, or is the launch bound by something that scales with the TILE COUNT
(HBM / fabric traffic)?  Times the four ViT-B/16 projection shapes on the pure persistent kernel (variant 18) at row-panel counts that give
2.0, 2.31 (the B = 256 batch), 2.5, 3.0 ... rounds of 256 x 256 tiles on 256 CUs.  If time follows ceil(rounds), the tail is idle CUs; if it
follows the tile count, the launch is bandwidth-bound and filling the tail with other work cannot be free.

    python tools/gemm_rounds_sweep.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)

    def rnd(*shape, dtype=torch.bfloat16, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev).to(dtype)

    shapes = [("qkv", 2304, 768, 0, False), ("out+res", 768, 768, 0, True), ("up+gelu", 3072, 768, 1, False), ("down+res", 768, 3072, 0, True)]
    ops.set_gemm_variant(18)
    for name, N, K, act, res in shapes:
        tn = N // 256
        w, bias = rnd(N, K, scale=0.05), rnd(N, dtype=torch.float32)
        row = []
        for rounds in (1.0, 2.0, 2.31, 2.5, 3.0, 4.0, 6.93, 7.0, 9.23, 10.0):
            tm = max(1, round(rounds * 256 / tn))
            M = tm * 256
            if M > 70000:
                continue
            a = rnd(M, K)
            out = torch.zeros((M, N), dtype=torch.float32 if res else torch.bfloat16, device=dev)
            ms = sorted(timeit(lambda: ops.gemm_bf16(a, w, bias, act=act, residual=out if res else None, out=out), 10) for _ in range(3))[1]
            tiles = tm * tn
            row.append(f"{tiles / 256:5.2f}r {ms * 1e3:6.1f}us ({ms * 1e3 / (tiles / 256):5.1f}/r)")
            del a, out
        print(f"{name:9s} " + " | ".join(row), flush=True)
    ops.set_gemm_variant(0)


if __name__ == "__main__":
    main()
