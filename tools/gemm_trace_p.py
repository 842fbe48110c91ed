#!/usr/bin/env python
"""Timeline of the pipelined GEMM (variant 164): prologue / per-K-tile / tail+epilogue cycles (waves 0 and 4, first 64 blocks)."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (50432, 2304, 768)
act = int(sys.argv[4]) if len(sys.argv) > 4 else 0
f32 = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
w = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16)
bias = torch.randn(N, generator=g).to(dev)
out = torch.zeros((M, N), dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
trace = torch.zeros(64 * 2 * 256, dtype=torch.int64, device=dev)
_lib.lib().mmamd_debug_set_gemm_trace(trace.data_ptr())
ops.set_gemm_variant(164)
for _ in range(3):
    ops.gemm_bf16(a, w, bias, act=act, residual=out if f32 else None, out=out)
torch.cuda.synchronize()
_lib.lib().mmamd_debug_set_gemm_trace(None)
t = trace.cpu().numpy().reshape(64, 2, 256)
for grp in (0, 1):
    rows = []
    for b in range(64):
        n = int(t[b, grp, 0])
        rows.append(np.diff(t[b, grp, 1:1 + n].astype(np.int64)))
    n = min(len(r) for r in rows)
    d = np.stack([r[:n] for r in rows]).mean(0)
    print(f"waves {'0-3' if grp == 0 else '4-7'}: {n} intervals;  prologue(first tile landed) {d[0]:.0f}  mean K-tile {d[1:n-3].mean():.0f}")
    print(f"   last tile + flush {d[n-3]:.0f}   epilogue to last store ISSUED {d[n-2]:.0f}   store drain {d[n-1]:.0f}   block total {d.sum():.0f} ticks")
