"""A few launches of FLAVA's attention with probabilities (image-encoder shape) for PMC collection:  python tools/one_probs.py [variant [B S H reps]]
variant = mmamd_debug_set_attn_variant value: 515 = flash forward + one-pass probabilities kernel (default), 514 = the two-pass kernel"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 515
B, S, H, reps = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (256, 197, 12, 5)))
_lib.lib().mmamd_debug_set_attn_variant(variant)
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
out = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device="cuda")
for _ in range(reps):
    ops.attention_probs_fwd(qkv, B, S, H, None, True, torch.float32, out=out)
torch.cuda.synchronize()
