"""mmamd_colsum on few rows of a very wide matrix (the positional-embedding gradient of the CLIP / FLAVA image towers: [B, S * w]): the r05 form (a thread
per 16-byte column chunk, eight row groups) against the row-per-workgroup form, alternating in one process.    python tools/colsum_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

L = _lib.lib()
for rows, n, dt in ((256, 197 * 768, torch.float32), (128, 197 * 768, torch.float32), (256, 50 * 768, torch.float32), (256, 257 * 1024, torch.float32)):
    x = torch.randn(rows, n, device="cuda").to(dt)
    res = {0: [], 1: []}
    outs = {}
    for rnd in range(3):
        for form in (0, 1):
            L.mmamd_debug_set_colsum_wide(form)
            res[form].append(timeit(lambda: ops.colsum(x), 30) * 1e3)
            outs[form] = ops.colsum(x)
    L.mmamd_debug_set_colsum_wide(1)
    mb = x.numel() * x.element_size() / 1e6
    err = (outs[0].double() - outs[1].double()).abs().max().item()
    print(f"[{rows}, {n}] {str(dt)[6:]}: {mb:.0f} MB; row-per-workgroup " + " ".join(f"{t:.1f}" for t in res[0]) + " us | wide " + " ".join(f"{t:.1f}" for t in res[1]) +
          f" us ({mb / min(res[1]):.2f} TB/s); max |difference| {err:.2e}", flush=True)
