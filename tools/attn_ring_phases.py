"""In-kernel phase timers of the ring attention kernel (build with MMAMD_EXPERIMENTS=1): s_memtime ticks per wave, averaged over workgroups.
python tools/attn_ring_phases.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402

L = _lib.lib()
B, S, H = 256, 197, 12
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
L.mmamd_debug_set_attn_variant(2000 + (int(sys.argv[1]) if len(sys.argv) > 1 else 4))
for _ in range(3):
    out, lse = ops.attention_fwd_train(qkv, B, S, H, False)
torch.cuda.synchronize()
t = lse.flatten()[: 256 * 8 * 8].reshape(256, 8, 8)[:, :, :4].double().cpu()
L.mmamd_debug_set_attn_variant(2000)
names_c = ("key loops", "O transpose + store", "barrier wait", "total")
names_l = ("DMA issue", "landing wait", "barrier wait", "total")
for w in range(8):
    m = t[:, w, :].mean(0)
    names = names_l if w == 7 else names_c
    tot = m[3].item()
    print(f"wave {w}: " + "  ".join(f"{n} {v.item() / 1e3:8.1f}k ({100 * v.item() / tot:4.1f} %)" for n, v in zip(names, m)))
