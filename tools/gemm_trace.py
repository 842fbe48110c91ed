#!/usr/bin/env python
"""Section-level timeline of the staggered GEMM (variant 14): mean cycles of LOAD / barrier / MATRIX / barrier."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (50432, 2304, 768)
VAR = int(sys.argv[4]) if len(sys.argv) > 4 else 14
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
w = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
trace = torch.zeros(64 * 2 * 256, dtype=torch.int64, device=dev)
_lib.lib().mmamd_debug_set_gemm_trace(trace.data_ptr())
ops.set_gemm_variant(VAR)
for _ in range(3):
    ops.gemm_bf16(a, w, out=out)
torch.cuda.synchronize()
_lib.lib().mmamd_debug_set_gemm_trace(None)
t = trace.cpu().numpy().reshape(64, 2, 256)
for grp in (0, 1):
    rows = []
    for b in range(64):
        n = int(t[b, grp, 0])
        st = t[b, grp, 1:1 + n].astype(np.int64)
        rows.append(np.diff(st))
    n = min(len(r) for r in rows)
    d = np.stack([r[:n] for r in rows])
    mean = d.mean(0)
    print(f"waves {'0-3' if grp == 0 else '4-7'}: stamps={n + 1}")
    print("  prologue (issue 3 stages + wait + barrier[+stagger]):", round(mean[0]))
    body = mean[1:n - 1]
    k = (len(body) // 7) * 7
    sec = body[:k].reshape(-1, 7)
    print("  per stage [DMA issue, ds_read issue, lgkm wait, vmcnt wait, barrier, MATRIX, barrier] mean cycles:", np.round(sec.mean(0)).tolist())
    print("  first 6 stages:", np.round(sec[:6]).tolist())
    print("  stage total:", round(sec.sum(1).mean()), "cycles; x", len(sec), "stages")
    print("  tail (drain stages, epilogue):", np.round(body[k:]).tolist(), round(mean[n - 1]))
    print("  whole block:", round(d.sum(1).mean()), "cycles")

