"""Does a deeper LDS ring pay?  Shapes whose ring entry is small enough for 3-4 entries in 160 KiB, timed with the depth capped at 2 / 3 / 4
(HIP-graph replays).   python tools/attn_ring_depth.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402
from tools.attn_ring_ablate import graph_time  # noqa: E402

L = _lib.lib()


def main():
    for (B, S, H, c) in ((256, 128, 12, False), (256, 112, 12, False), (256, 100, 12, False), (512, 77, 8, True), (256, 197, 12, False)):
        torch.manual_seed(0)
        qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
        out = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device="cuda")
        nqt, rows8 = (S + 31) // 32, (S + 7) & ~7
        entry = (7 // nqt) * (2 * rows8 + 32 * nqt) * 128
        mb = (B * S * 4 * H * 64) * 2 / 1e6
        for rnd in range(2):
            for cap in (2, 3, 4):
                if cap > max(2, (160 * 1024) // entry):
                    continue
                L.mmamd_debug_set_attn_variant(3000 + cap)
                us = graph_time(lambda: ops.attention_fwd(qkv, B, S, H, c, out=out))
                print(f"B={B} S={S} H={H} causal={int(c)} entry={entry / 1024:.1f} KiB depth<={cap}: {us:7.1f} us  {mb / us:5.2f} TB/s", flush=True)
        L.mmamd_debug_set_attn_variant(3000)




def noq():
    """S = 197 with the Q rows taken out of the ring entry (results wrong, timing only; MMAMD_EXPERIMENTS build): does a third entry pay when
    compute and memory are both critical?   python tools/attn_ring_depth.py noq"""
    B, S, H, c = 256, 197, 12, False
    torch.manual_seed(0)
    qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
    out = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device="cuda")
    for rnd in range(3):
        for abl, tag in ((402, "Q in ring, no O stores"), (400, "no Q rows, no O stores"), (401, "no Q rows, O stores"), (0, "shipped")):
            for cap in (2, 3):
                if cap == 3 and abl in (402, 0):
                    continue
                L.mmamd_debug_set_attn_variant(3000 + cap)
                L.mmamd_debug_set_attn_variant(2000 + abl)
                us = graph_time(lambda: ops.attention_fwd(qkv, B, S, H, c, out=out))
                print(f"{tag:26s} depth<={cap}: {us:7.1f} us", flush=True)
    L.mmamd_debug_set_attn_variant(2000)
    L.mmamd_debug_set_attn_variant(3000)


def store_policy():
    B, S, H, c = 256, 197, 12, False
    torch.manual_seed(0)
    qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
    out = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device="cuda")
    ref = None
    for rnd in range(3):
        for abl, tag in ((0, "plain O stores"), (410, "nt O stores"), (411, "sc1 O stores")):
            L.mmamd_debug_set_attn_variant(2000 + abl)
            us = graph_time(lambda: ops.attention_fwd(qkv, B, S, H, c, out=out))
            ref = out.clone() if ref is None else ref
            print(f"{tag:26s}: {us:7.1f} us   equal={torch.equal(out, ref)}", flush=True)
    L.mmamd_debug_set_attn_variant(2000)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "noq":
    noq()
elif __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "store":
    store_policy()
elif __name__ == "__main__":
    main()
