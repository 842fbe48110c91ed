"""A/B of the probabilities-from-lse kernel with 8 vs 16 waves per workgroup (mmamd_debug_set_attn_variant(5308 / 5316)), the kernel alone through
mmamd_attention_probs_from_lse, alternating arms.   python tools/probs_lse_waves_ab.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402
from tools.probs_lse_bench import timed  # noqa: E402


def main():
    L = _lib.lib()
    for B, S, H in ((256, 197, 12), (128, 275, 12), (256, 129, 12)):
        torch.manual_seed(0)
        qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
        _, lse = ops.attention_fwd_train(qkv, B, S, H, False)
        res = {8: [], 16: []}
        ref = None
        for rnd in range(3):
            for nw in (8, 16):
                L.mmamd_debug_set_attn_variant(5300 + nw)
                res[nw].append(timed(lambda: ops.attention_probs_from_lse(qkv, lse, B, S, H)))
                p = ops.attention_probs_from_lse(qkv, lse, B, S, H)
                ref = p if ref is None else ref
                assert torch.equal(p, ref), "the two forms must store the same values"
        print(f"B={B} S={S}: 8 waves " + " ".join(f"{t:6.1f}" for t in res[8]) + " us | 16 waves " + " ".join(f"{t:6.1f}" for t in res[16]) + " us", flush=True)
    L.mmamd_debug_set_attn_variant(5300)  # back to the by-length default


if __name__ == "__main__":
    main()
