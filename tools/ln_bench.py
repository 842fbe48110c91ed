"""LayerNorm launches of one ViT-B/16 + text layer (B = 256), graph-timed: two separate mmamd_layernorm launches (r02) vs the grouped
launch, and the grouped residual-add + LayerNorm launch of the delta_ln schedule.   python tools/ln_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402
from tools.attn_ring_ablate import graph_time  # noqa: E402

dev = "cuda"
Ma, da, Mb, db = 256 * 197, 768, 256 * 77, 512
torch.manual_seed(0)
xa, xb = torch.randn(Ma, da, device=dev), torch.randn(Mb, db, device=dev)
dla, dlb = torch.randn(Ma, da, device=dev).bfloat16(), torch.randn(Mb, db, device=dev).bfloat16()
ga, ba, gb, bb = torch.randn(da, device=dev), torch.randn(da, device=dev), torch.randn(db, device=dev), torch.randn(db, device=dev)
ya, yb = torch.empty(Ma, da, device=dev, dtype=torch.bfloat16), torch.empty(Mb, db, device=dev, dtype=torch.bfloat16)


def two():
    ops.layernorm(xa, ga, ba, 1e-5, out=ya)
    ops.layernorm(xb, gb, bb, 1e-5, out=yb)


for rnd in range(3):
    t2 = graph_time(two)
    tg = graph_time(lambda: ops.add_layernorm_grouped([(xa, None, ga, ba, 1e-5, ya), (xb, None, gb, bb, 1e-5, yb)]))
    t1 = graph_time(lambda: ops.add_layernorm_grouped([(xa, None, ga, ba, 1e-5, ya)]))
    td = graph_time(lambda: ops.add_layernorm_grouped([(xa, dla, ga, ba, 1e-5, ya), (xb, dlb, gb, bb, 1e-5, yb)]))
    mb_ln = (Ma * da * 6 + Mb * db * 6) / 1e6
    mb_add = (Ma * da * 12 + Mb * db * 12) / 1e6
    print(f"two launches {t2:6.1f} us | grouped {tg:6.1f} us ({mb_ln / tg:4.2f} TB/s) | ViT rows alone {t1:6.1f} us | grouped add+LN {td:6.1f} us ({mb_add / td:4.2f} TB/s)", flush=True)
