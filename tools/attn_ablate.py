#!/usr/bin/env python
"""Attention forward ablations / schedule variants on the vision shape (experiment build: MMAMD_EXPERIMENTS=1), warm-clock timing."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

B, S, H = 256, 197, 12
dev = torch.device("cuda", 0)
qkv = torch.randn(B * S, 3 * H * 64).to(dev).to(torch.bfloat16)
o = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device=dev)
for v, name in ((0, "default: software-pipelined key loop, V by transpose reads"), (128, "serial key loop"), (1, "V staged row-major, plain reads (WRONG results)"),
                (2, "no exp (WRONG)"), (3, "1+2"), (4, "no K/V global loads (WRONG)"), (7, "1+2+4"), (0, "default again")):
    _lib.lib().mmamd_debug_set_attn_variant(v)
    ms = timeit(lambda: ops.attention_fwd(qkv, B, S, H, False, out=o), 50)
    print(f"attn variant {v:3d} ({name}): {ms * 1e3:.1f} us", flush=True)
# layout experiment: the same q / k / v HEAD-major ([3H][B*S][64]: every (batch, head) slice contiguous); results must equal the default's
hm = qkv.view(B * S, 3 * H, 64).permute(1, 0, 2).contiguous().view(B * S, 3 * H * 64)
ref = ops.attention_fwd(qkv, B, S, H, False).clone()
_lib.lib().mmamd_debug_set_attn_variant(256)
got = ops.attention_fwd(hm, B, S, H, False, out=o)
print("head-major input: max |d out| vs default layout =", float((got.float() - ref.float()).abs().max()))
ms = timeit(lambda: ops.attention_fwd(hm, B, S, H, False, out=o), 50)
print(f"attn variant 256 (head-major qkv): {ms * 1e3:.1f} us", flush=True)
_lib.lib().mmamd_debug_set_attn_variant(0)
ms = timeit(lambda: ops.attention_fwd(qkv, B, S, H, False, out=o), 50)
print(f"attn variant   0 (default, after): {ms * 1e3:.1f} us", flush=True)
