"""A/B of the LDS-DMA ring attention kernel (default) against the r02 register-staged kernel (mmamd_debug_set_attn_variant(1000)),
interleaved rounds in one process (guide rule 24), warm clocks (tools/kernel_bench.py::timeit), plus the grouped ViT + text launch
against the two separate launches.   python tools/attn_ring_bench.py [--rounds 5] [--batch 256]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
B = args.batch
L = _lib.lib()
shapes = (("vit-b16", B, 197, 12, False), ("text-77", B, 77, 8, True), ("vit-b32", B, 50, 12, False), ("flava-txt", B, 128, 12, False))
bufs = {}
for name, b, S, H, causal in shapes:
    torch.manual_seed(0)
    qkv = torch.randn(b * S, 3 * H * 64).to(torch.bfloat16).cuda()
    out = torch.empty((b * S, H * 64), dtype=torch.bfloat16, device="cuda")
    bufs[name] = (qkv, out)
res = {}
for rnd in range(args.rounds):
    for name, b, S, H, causal in shapes:
        qkv, out = bufs[name]
        for var, tag in ((0, "ring"), (1000, "r02")):
            L.mmamd_debug_set_attn_variant(var)
            ms = timeit(lambda: ops.attention_fwd(qkv, b, S, H, causal, out=out), 20)
            res.setdefault((name, tag), []).append(ms * 1e3)
    L.mmamd_debug_set_attn_variant(0)
    (qa, oa), (qb, ob) = bufs["vit-b16"], bufs["text-77"]
    ms = timeit(lambda: ops.attention_fwd_grouped([(qa, B, 197, 12, False, oa), (qb, B, 77, 8, True, ob)]), 20)
    res.setdefault(("vit+text", "grouped"), []).append(ms * 1e3)

    def two():
        ops.attention_fwd(qa, B, 197, 12, False, out=oa)
        ops.attention_fwd(qb, B, 77, 8, True, out=ob)

    ms = timeit(two, 20)
    res.setdefault(("vit+text", "2 launches"), []).append(ms * 1e3)
L.mmamd_debug_set_attn_variant(0)
for (name, tag), v in res.items():
    v = sorted(v)
    print(f"{name:10s} {tag:11s} median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f}  max {v[-1]:7.1f}   ({len(v)} rounds)", flush=True)
for name, b, S, H, causal in shapes:
    mb = (b * S * 3 * H * 64 + b * S * H * 64) * 2 / 1e6
    v = sorted(res[(name, "ring")])
    print(f"{name:10s} algorithmic {mb:6.1f} MB -> {mb / v[len(v) // 2]:5.2f} TB/s (ring, median) = {mb / v[len(v) // 2] / 8.0:4.2f} of 8 TB/s")
