#!/usr/bin/env python
"""What the second tower's tiles cost inside a grouped persistent launch: every projection pair of the cfg-2 layer timed as ViT only, text only
and grouped (ViT + text) under the walk orders / stagger policies of the grouped kernel (mmamd_debug_set_gemm_knob), interleaved rounds, median.

    python tools/grouped_order_bench.py [--rounds 3] [--batch 256]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--arms", default="order0,order1,order2,stagger0,slack60")
    a = ap.parse_args()
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)

    def rnd(*shape, dtype=torch.bfloat16, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev).to(dtype)

    Mv, Mt = a.batch * 197, a.batch * 77
    pairs = [("qkv", (Mv, 2304, 768), (Mt, 1536, 512), 0, False), ("out+res", (Mv, 768, 768), (Mt, 512, 512), 0, True),
             ("up+gelu", (Mv, 3072, 768), (Mt, 2048, 512), 1, False), ("down+res", (Mv, 768, 3072), (Mt, 512, 2048), 0, True)]

    def set_arm(name):
        L.mmamd_debug_set_gemm_stagger(60)
        for k in (0, 1, 2):
            L.mmamd_debug_set_gemm_knob(k, 0)
        for part in name.split("+"):
            if part.startswith("order"):
                L.mmamd_debug_set_gemm_knob(2, int(part[5:]))
            elif part.startswith("stagger"):
                L.mmamd_debug_set_gemm_stagger(int(part[7:]))
            elif part.startswith("slack"):
                L.mmamd_debug_set_gemm_knob(1, int(part[5:]))
            elif part.startswith("gm"):
                L.mmamd_debug_set_gemm_knob(0, int(part[2:]))
            else:
                raise SystemExit(f"unknown arm {part}")

    arms = a.arms.split(",")
    for name, pv, pt, act, res in pairs:
        def prob(M, N, K):
            out = torch.zeros((M, N), dtype=torch.float32 if res else torch.bfloat16, device=dev)
            return (rnd(M, K), rnd(N, K, scale=0.05), rnd(N, dtype=torch.float32), out if res else None, out)

        v, t = prob(*pv), prob(*pt)
        odt = torch.float32 if res else torch.bfloat16
        res_ms = {}
        for _ in range(a.rounds):
            set_arm("order0")
            res_ms.setdefault("vit_only", []).append(timeit(lambda: ops.gemm_bf16_grouped([v], act=act, out_dtype=odt), 10))
            res_ms.setdefault("text_only", []).append(timeit(lambda: ops.gemm_bf16_grouped([t], act=act, out_dtype=odt), 10))
            for arm in arms:
                set_arm(arm)
                res_ms.setdefault(arm, []).append(timeit(lambda: ops.gemm_bf16_grouped([v, t], act=act, out_dtype=odt), 10))
        set_arm("order0")
        med = {k: sorted(x)[len(x) // 2] * 1e3 for k, x in res_ms.items()}
        print(f"{name:9s} " + "  ".join(f"{k} {us:7.1f}" for k, us in med.items()), flush=True)


if __name__ == "__main__":
    main()
