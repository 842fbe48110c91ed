"""Times the attention forward kernels (HIP events, in-order stream):  python tools/attn_bench.py [--iters N]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
args = ap.parse_args()
for name, B, S, H, causal in (("vit-b16", 256, 197, 12, False), ("vit-l14", 256, 257, 16, False), ("vit-b32", 256, 50, 12, False),
                              ("text-77", 256, 77, 8, True), ("flava-txt", 256, 128, 12, False)):
    torch.manual_seed(0)
    qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
    out = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device="cuda")
    for _ in range(5):
        ops.attention_fwd(qkv, B, S, H, causal, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        ops.attention_fwd(qkv, B, S, H, causal, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.iters * 1e3
    flops = 4.0 * B * H * S * S * 64 * (0.5 if causal else 1.0)
    print(f"{name:10s} B={B} S={S} H={H} causal={int(causal)}  {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
