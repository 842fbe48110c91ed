#!/usr/bin/env python
"""The ViT-B/16 stem at B = 256, piece by piece (HIP-graph replays of each launch, a 310 MB fill in front of each so nothing is cache-warm):
fused form (fp32 -> bf16 cast, patch-embedding GEMM that gathers the patches in its DMA addresses, CLS + ln_pre + norm1 row kernel) against the
patch-matrix form (patchify, plain GEMM, assemble + ln_pre).   python tools/stem_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402


def timed(fn, scratch, iters=20):
    ts = []
    for i in range(iters + 3):
        scratch.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    B, P, HW, W = 256, 16, 224, 768
    g = HW // P
    img = torch.randn(B, 3, HW, HW, device=dev)
    w = (torch.randn(W, 3 * P * P, device=dev) * 0.02).to(torch.bfloat16)
    pos = torch.randn(g * g + 1, W, device=dev) * 0.02
    cls = torch.randn(W, device=dev) * 0.02
    gam, bet = torch.ones(W, device=dev), torch.zeros(W, device=dev)
    scratch = torch.empty(B * 197, 3072, dtype=torch.bfloat16, device=dev)
    empty = timed(lambda: None, scratch)
    img16 = ops.convert(img, torch.bfloat16)
    h = ops.patch_embed_fused(img16, w, pos, P)
    patches = ops.patchify(img, P, 768)
    pe = ops.gemm_bf16(patches, w, out_dtype=torch.float32)
    rows = [("(empty: events + the fill's tail)", lambda: None),
            ("convert fp32 -> bf16 image", lambda: ops.convert(img, torch.bfloat16, out=img16)),
            ("patch-embedding GEMM, gathered A (+ pos)", lambda: ops.patch_embed_fused(img16, w, pos, P)),
            ("CLS + ln_pre + norm1 rows", lambda: ops.vit_cls_lnpre_ln(h, cls, pos, gam, bet, 1e-5, B, g * g + 1, (gam, bet, 1e-5))),
            ("patchify fp32 image -> bf16 patch rows", lambda: ops.patchify(img, P, 768)),
            ("plain GEMM on the patch rows (fp32 out)", lambda: ops.gemm_bf16(patches, w, out_dtype=torch.float32, out=pe)),
            ("assemble + pos + ln_pre rows", lambda: ops.vit_assemble_ln(pe, cls, pos, gam, bet, 1e-5, B, g * g))]
    for name, fn in rows:
        print(f"{name:46s} {timed(fn, scratch) - (0 if fn.__name__ == '<lambda>' and name.startswith('(empty') else empty):7.1f} us", flush=True)


if __name__ == "__main__":
    main()
