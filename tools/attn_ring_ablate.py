"""Ablations / A-B of the attention kernels timed as HIP-graph replays (20 launches per graph: the Python + ctypes call costs ~90 us,
more than the kernel, so eager loops measure the host).   python tools/attn_ring_ablate.py"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402

L = _lib.lib()
NL = 20


def graph_time(fn, reps=30):
    """us per launch of fn (captured NL times into one graph, replayed reps times after a warm-up)"""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(NL):
                fn()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * NL)


def main():
    shapes = {"vit-b16": (256, 197, 12, False), "text-77": (256, 77, 8, True), "vit-b32": (256, 50, 12, False), "flava-txt": (256, 128, 12, False)}
    bufs = {}
    for name, (B, S, H, c) in shapes.items():
        torch.manual_seed(0)
        bufs[name] = (torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda(), torch.empty((B * S, H * 64), dtype=torch.bfloat16, device="cuda"))
    for rnd in range(0 if os.environ.get("ABLS") else 2):
        for name, (B, S, H, c) in shapes.items():
            qkv, out = bufs[name]
            for var, tag in ((0, "ring"), (1000, "r02")):
                L.mmamd_debug_set_attn_variant(var)
                us = graph_time(lambda: ops.attention_fwd(qkv, B, S, H, c, out=out))
                mb = (B * S * 4 * H * 64) * 2 / 1e6
                print(f"{name:10s} {tag:5s} {us:7.1f} us   {mb / us:5.2f} TB/s algorithmic", flush=True)
        L.mmamd_debug_set_attn_variant(0)
        (qa, oa), (qb, ob) = bufs["vit-b16"], bufs["text-77"]
        us = graph_time(lambda: ops.attention_fwd_grouped([(qa, 256, 197, 12, False, oa), (qb, 256, 77, 8, True, ob)]))
        print(f"vit+text   grouped {us:7.1f} us", flush=True)
    B, S, H, c = shapes["vit-b16"]
    qkv, out = bufs["vit-b16"]
    abls = [int(x) for x in os.environ.get("ABLS", "0,16,9,25,1,17,2,8,0,16").split(",")]
    for abl in abls:
        L.mmamd_debug_set_attn_variant(2000 + abl)
        us = graph_time(lambda: ops.attention_fwd(qkv, B, S, H, c, out=out))
        print(f"ring abl={abl:3d} (noDMA={abl & 1} noKeyLoop={(abl >> 1) & 1} noOstore={(abl >> 3) & 1} runtimeLoop={(abl >> 4) & 1} noExp={(abl >> 5) & 1} "
              f"noLdsReads={(abl >> 6) & 1} noMfma={(abl >> 7) & 1} noXchg={(abl >> 8) & 1}): {us:7.1f} us", flush=True)
    L.mmamd_debug_set_attn_variant(2000)


if __name__ == "__main__":
    main()
