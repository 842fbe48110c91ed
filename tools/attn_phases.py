#!/usr/bin/env python
"""Phase timers of the persistent attention forward (vision shape): s_memtime cycles per wave, summed over its items.
python tools/attn_phases.py   — variants: 8 = pipelined key loop, 136 = serial key loop"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

B, S, H = 256, 197, 12
dev = torch.device("cuda", 0)
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * H * 64).to(dev).to(torch.bfloat16)
o = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device=dev)
L = _lib.lib()
for v, name in ((0, "default"), (128, "serial key loop"), (8, "default + timers"), (136, "serial + timers")):
    L.mmamd_debug_set_attn_variant(v)
    ms = timeit(lambda: ops.attention_fwd(qkv, B, S, H, False, out=o), 20)
    print(f"variant {v:3d} ({name}): {ms * 1e3:.1f} us")
# clocks ramp? same kernels after ~1 s of dense GEMM work and over a long loop
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
for _ in range(300):
    a @ a
torch.cuda.synchronize()
for v, name in ((0, "default"), (128, "serial key loop")):
    L.mmamd_debug_set_attn_variant(v)
    ms = timeit(lambda: ops.attention_fwd(qkv, B, S, H, False, out=o), 2000)
    print(f"after GEMM warm-up, 2000 iters: variant {v:3d} ({name}): {ms * 1e3:.1f} us")
names = ["store_item", "barrier1", "key loops", "epilogue+Q wait", "barrier2", "total"]
for v, name in ((8, "pipelined"), (136, "serial")):
    L.mmamd_debug_set_attn_variant(v)
    buf = torch.zeros(B * H * S, dtype=torch.float32, device=dev)  # lse-sized: >= 512*4*8 floats
    for _ in range(3):
        ops.check(L.mmamd_attention_fwd_lse(qkv.data_ptr(), o.data_ptr(), buf.data_ptr(), B, S, H, 0, 0.125, torch.cuda.current_stream().cuda_stream), "attention_fwd_lse")
    torch.cuda.synchronize()
    t = buf[: 512 * 4 * 8].view(512, 4, 8).cpu()
    print(f"--- {name}: mean s_memtime ticks per wave (100 MHz ticks x ~21-24 = shader cycles)")
    for w in range(4):
        print(f"  wave {w}: " + "  ".join(f"{n}={t[:, w, i].mean():9.0f}" for i, n in enumerate(names)))
L.mmamd_debug_set_attn_variant(0)
