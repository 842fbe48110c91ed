#!/usr/bin/env python
"""Isolated timings (warm clocks) + bit-equality of GEMM kernel variants on the cfg-2 problem shapes:
    python tools/gemm_variant_bench.py --variants 0,50,51   (0 = the default dispatch policy)"""
import argparse
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tools.kernel_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,50,51,52")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--only", default="", help="comma-separated shape-name substrings")
    ap.add_argument("--set", default="clip", choices=["clip", "flava", "coca"], help="problem shapes: the cfg-2 CLIP step (default), FLAVA cfg 4, CoCa cfg 5")
    ap.add_argument("--staggers", default="", help="comma-separated stagger tick counts: sweeps them on variant 0 instead of the variants")
    args = ap.parse_args()
    from multimodal_amd import build, ops

    build.build()
    variants = [int(v) for v in args.variants.split(",")]
    g = torch.Generator().manual_seed(0)

    def rnd(*shape, dtype=torch.bfloat16, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).cuda().to(dtype)

    shapes = [("v.qkv", 50432, 2304, 768, ops.ACT_NONE, False), ("v.mlp_up", 50432, 3072, 768, ops.ACT_QUICKGELU, False),
              ("v.out_proj", 50432, 768, 768, ops.ACT_NONE, True), ("v.mlp_down", 50432, 768, 3072, ops.ACT_NONE, True),
              ("t.qkv", 19712, 1536, 512, ops.ACT_NONE, False), ("t.mlp_up", 19712, 2048, 512, ops.ACT_QUICKGELU, False),
              ("t.out_proj", 19712, 512, 512, ops.ACT_NONE, True), ("t.mlp_down", 19712, 512, 2048, ops.ACT_NONE, True),
              ("patch", 50176, 768, 768, ops.ACT_NONE, None)]
    if args.set == "flava":  # cfg 4, B = 128: image tower 197 tokens, text 77, fusion 275 (d = 768, erf-GELU MLPs)
        shapes = [(f"{t}.{n}", M, N, K, act, res) for t, M in (("i", 25216), ("t", 9856), ("m", 35200))
                  for n, N, K, act, res in (("qkv", 2304, 768, ops.ACT_NONE, False), ("out", 768, 768, ops.ACT_NONE, True),
                                            ("up", 3072, 768, ops.ACT_GELU_ERF, False), ("down", 768, 3072, ops.ACT_NONE, True))]
    elif args.set == "coca":  # cfg 5 per-GPU shape, B = 128: ViT-L/14 256 tokens (d = 1024), text / fusion decoders 77 tokens (d = 768)
        shapes = [(f"{t}.{n}", M, N, K, act, res) for t, M, d in (("v", 32768, 1024), ("t", 9856, 768))
                  for n, N, K, act, res in (("qkv", 3 * d, d, ops.ACT_NONE, False), ("out", d, d, ops.ACT_NONE, True),
                                            ("up", 4 * d, d, ops.ACT_GELU_ERF, False), ("down", d, 4 * d, ops.ACT_NONE, True))]
    for name, M, N, K, act, res in shapes:
        if args.only and not any(t in name for t in args.only.split(",")):
            continue
        a, w, bias = rnd(M, K), rnd(N, K, scale=0.05), rnd(N, dtype=torch.float32)
        f32out = res is not False
        x0 = rnd(M, N, dtype=torch.float32) if res else None
        ref = None
        line = f"{name:10s} [{M}x{N}x{K}]"
        sweep = [(0, int(x)) for x in args.staggers.split(",")] if args.staggers else [(v, 0) for v in variants]
        wp = ops.pack_w_frag(w)  # fragment-order W of the direct-W variants (84 / 85); ignored by the others
        ops.debug_set_gemm_wp(wp)
        for v, stg in sweep:
            ops.set_gemm_variant(v)
            if args.staggers:
                ops.set_gemm_stagger(stg)
            out = x0.clone() if res else torch.empty(M, N, dtype=torch.float32 if f32out else torch.bfloat16, device="cuda")

            def run():
                # residual GEMMs update x in place, as the transformer stack does (the values drift over the timing loop; only the first call is compared)
                ops.gemm_bf16(a, w, bias, act=act, residual=out if res else None, out=out)

            run()
            torch.cuda.synchronize()
            if res:
                out_first = out.clone()
            else:
                out_first = out
            if ref is None:
                ref = out_first.clone()
                same = "ref"
            else:
                same = "==" if torch.equal(out_first, ref) else f"!= (max {float((out_first.float() - ref.float()).abs().max()):.3g})"
            best = min(timeit(run, 10) for _ in range(args.rounds))
            line += f" | v{v}{'/s' + str(stg) if stg else ''}: {best * 1e3:7.1f} us {2.0 * M * N * K / best / 1e9:5.0f} TF/s {same}"
        ops.set_gemm_variant(0)
        ops.debug_set_gemm_wp(None)
        if args.staggers:
            ops.set_gemm_stagger(60)  # the default
        print(line, flush=True)


if __name__ == "__main__":
    main()
