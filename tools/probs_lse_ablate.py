"""Where the time of the probabilities-from-lse kernel (csrc/attention_probs_lse.hip) goes: ablation bits of its S = 197 instantiation and an occupancy
A/B, timed alone through mmamd_attention_probs_from_lse (needs a build with MMAMD_EXPERIMENTS=1).   python tools/probs_lse_ablate.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


def main():
    L = _lib.lib()
    B, S, H = 256, 197, 12
    torch.manual_seed(0)
    qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
    _, lse = ops.attention_fwd_train(qkv, B, S, H, False)
    names = {0: "as built", 1: "no global stores", 2: "no compute phase", 4: "no LDS reads in the streaming phase", 3: "no stores, no compute",
             5: "no stores, no streaming reads", 6: "stores only (no compute, no streaming reads)", 8: "no barrier", 14: "stores only, no barrier"}
    for rnd in range(2):
        for abl in (0, 1, 2, 4, 3, 5, 6, 8, 14):
            L.mmamd_debug_set_attn_variant(5000 + abl)
            t = timed(lambda: ops.attention_probs_from_lse(qkv, lse, B, S, H))
            print(f"abl {abl:2d}  {names[abl]:48s} {t:7.1f} us", flush=True)
    L.mmamd_debug_set_attn_variant(5000)
    for pad, per_cu in ((0, "two"), (10, "one")):
        L.mmamd_debug_set_attn_variant(5100 + pad)
        t = timed(lambda: ops.attention_probs_from_lse(qkv, lse, B, S, H))
        print(f"extra LDS {pad:3d} KiB per workgroup ({per_cu} per CU)        {t:7.1f} us", flush=True)
    L.mmamd_debug_set_attn_variant(5100)
    fill = torch.empty((B, H, S, S), dtype=torch.float32, device="cuda")
    print(f"fill_ of the same bytes                                   {timed(lambda: fill.fill_(0.5)):7.1f} us")


if __name__ == "__main__":
    main()
