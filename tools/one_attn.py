"""A few launches of the vision-shape attention forward, for PMC collection:  python tools/one_attn.py [B S H causal reps]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402

B, S, H, causal, reps = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (256, 197, 12, 0, 5)))
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
out = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device="cuda")
for _ in range(reps):
    ops.attention_fwd(qkv, B, S, H, bool(causal), out=out)
torch.cuda.synchronize()
