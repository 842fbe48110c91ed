#!/usr/bin/env python
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops
from tools.kernel_bench import timeit
B, S, H = 256, 197, 12
dev = torch.device("cuda", 0)
qkv = torch.randn(B * S, 3 * H * 64).to(dev).to(torch.bfloat16)
o = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device=dev)
for v, name in ((0, "full"), (1, "V row-major (no b16 transposed stores)"), (2, "no exp"), (3, "1+2"), (4, "no K/V global loads"), (7, "1+2+4")):
    _lib.lib().mmamd_debug_set_attn_variant(v)
    ms = timeit(lambda: ops.attention_fwd(qkv, B, S, H, False, out=o), 20)
    print(f"attn variant {v} ({name}): {ms*1e3:.1f} us")
_lib.lib().mmamd_debug_set_attn_variant(0)
