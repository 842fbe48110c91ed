"""A few launches of the vision-shape attention forward, for PMC collection:  python tools/one_attn.py [variant [B S H causal reps]]
variant = mmamd_debug_set_attn_variant value (0 = ring kernel, 1000 = r02 register-staged kernel, 2000 + bits = ring ablations)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, S, H, causal, reps = (int(x) for x in (sys.argv[2:7] if len(sys.argv) > 6 else (256, 197, 12, 0, 5)))
_lib.lib().mmamd_debug_set_attn_variant(variant)
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
out = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device="cuda")
for _ in range(reps):
    ops.attention_fwd(qkv, B, S, H, bool(causal), out=out)
torch.cuda.synchronize()
