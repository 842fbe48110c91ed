#!/usr/bin/env python
"""Slot-level timeline of the ping-pong GEMM (variant 31, MMAMD_EXPERIMENTS build): mean s_memtime ticks of
[LOAD issue, barrier-1 wait, lgkmcnt wait, MFMA burst issue, barrier-2 wait] per phase (ticks are relative only)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (50432, 2304, 768)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
w = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16)
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
trace = torch.zeros(64 * 2 * 256, dtype=torch.int64, device=dev)
_lib.lib().mmamd_debug_set_gemm_trace(trace.data_ptr())
ops.set_gemm_variant(31)
for _ in range(3):
    ops.gemm_bf16(a, w, out=out)
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
    print(f"group {grp}: {n + 1} stamps; prologue(issue+wait) {d[0]:.0f}")
    body = d[1:]
    nph = (len(body) - 3) // 5
    sec = body[:nph * 5].reshape(nph, 5)  # [LOAD issue, bar1, lgkm, MFMA, bar2] x phases
    print("  phases:", nph, " mean per phase [LOAD, bar1, lgkm, MFMA, bar2]:", np.round(sec.mean(0)).tolist(), " phase total", round(sec.sum(1).mean()))
    for ph in range(4):
        print(f"   phase {ph}:", np.round(sec[ph::4].mean(0)).tolist())
    print("  tail [final bar/epilogue issue, store drain]:", np.round(body[nph * 5:]).tolist(), " whole block:", round(d.sum()))
