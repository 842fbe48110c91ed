"""attention_probs_fwd (FLAVA's attention with the [B,H,S,S] probabilities returned) over row lengths around S = 197: is the write of the
788-byte (unaligned) probability rows what holds it at 0.44 of the HBM peak?   python tools/attn_probs_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402


def main():
    B, H = 256, 12
    for S in (192, 197, 200, 208, 224):
        torch.manual_seed(0)
        qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
        res = {}
        for want, dt in ((True, torch.float32), (True, torch.bfloat16), (False, torch.float32)):
            for _ in range(3):
                ops.attention_probs_fwd(qkv, B, S, H, None, want, dt)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.attention_probs_fwd(qkv, B, S, H, None, want, dt)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            mb = (B * S * 4 * H * 64 * 2 + (B * H * S * S * (4 if dt == torch.float32 else 2) if want else 0)) / 1e6
            res[(want, dt)] = (us, mb / us)
        print(f"S={S:4d} row {S * 4:4d} B | fp32 probs {res[(True, torch.float32)][0]:7.1f} us {res[(True, torch.float32)][1]:5.2f} TB/s | bf16 probs "
              f"{res[(True, torch.bfloat16)][0]:7.1f} us {res[(True, torch.bfloat16)][1]:5.2f} TB/s | no probs {res[(False, torch.float32)][0]:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
