#!/usr/bin/env python
"""Isolated timings of the LN-fold GEMMs against the kernels they replace (one MI355X, warm clocks, HIP events on the launch stream):
    layernorm + gemm(+bias[,QuickGELU])      vs   gemm_bf16_lnfold          (consumer side: qkv, MLP-up)
    gemm(+bias, +fp32 residual)              vs   gemm_bf16_res_stats       (producer side: out-proj, MLP-down)
for the ViT-B/16 (50432 rows, d 768) and text (19712 rows, d 512) towers of cfg 2."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tools.kernel_bench import timeit  # noqa: E402


def main():
    from multimodal_amd import build, ops

    build.build()
    g = torch.Generator().manual_seed(0)

    def rnd(*shape, dtype=torch.bfloat16, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).cuda().to(dtype)

    for tower, M, d, ff in (("vision", 50432, 768, 3072), ("text", 19712, 512, 2048)):
        x = rnd(M, d, dtype=torch.float32)
        xh = torch.empty(M, d, dtype=torch.bfloat16, device="cuda")
        stats = torch.empty(M, d // 64, 2, device="cuda")
        ops.row_stats(x, xh, stats)
        gamma, beta = rnd(d, dtype=torch.float32), rnd(d, dtype=torch.float32)
        hn = torch.empty(M, d, dtype=torch.bfloat16, device="cuda")
        t_ln = timeit(lambda: ops.layernorm(x, gamma, beta, 1e-5, out=hn), 20)
        print(f"{tower:6s} layernorm [{M}x{d}]                       {t_ln * 1e3:8.1f} us")
        for name, N, act in (("qkv", 3 * d, ops.ACT_NONE), ("mlp_up", ff, ops.ACT_QUICKGELU)):
            w = rnd(N, d, dtype=torch.float32, scale=0.05)
            bias = rnd(N, dtype=torch.float32)
            wb = w.to(torch.bfloat16)
            out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            wg, c1, c2 = ops.lnfold_pack(w, gamma, beta, bias)
            t0 = timeit(lambda: ops.gemm_bf16(hn, wb, bias, act=act, out=out), 20)
            t1 = timeit(lambda: ops.gemm_bf16_lnfold(xh, wg, c1, c2, stats, 1e-5, act=act, out=out), 20)
            fl = 2.0 * M * N * d
            print(f"{tower:6s} {name:7s} [{M}x{N}x{d}]  plain {t0 * 1e3:7.1f} us ({fl / t0 / 1e9:6.0f} TF/s)   lnfold {t1 * 1e3:7.1f} us ({fl / t1 / 1e9:6.0f} TF/s)"
                  f"   ln+plain {(t0 + t_ln) * 1e3:7.1f}")
        for name, K in (("out_proj", d), ("mlp_down", ff)):
            a = rnd(M, K)
            w = rnd(d, K, scale=0.05)
            bias = rnd(d, dtype=torch.float32)
            t0 = timeit(lambda: ops.gemm_bf16(a, w, bias, residual=x, out=x), 20)
            t1 = timeit(lambda: ops.gemm_bf16_res_stats(a, w, bias, x, xh, stats), 20)
            fl = 2.0 * M * d * K
            print(f"{tower:6s} {name:8s} [{M}x{d}x{K}]  plain {t0 * 1e3:7.1f} us ({fl / t0 / 1e9:6.0f} TF/s)   res_stats {t1 * 1e3:7.1f} us ({fl / t1 / 1e9:6.0f} TF/s)")


if __name__ == "__main__":
    main()
