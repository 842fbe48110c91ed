#!/usr/bin/env python
"""Out-projection + residual + norm2 of one layer of both towers (CLIP ViT-B/16 + text, B = 256): the grouped GEMM + grouped LayerNorm of the shipped
schedule against the one-launch whole-row kernel (gemm_rowln.hip).   python tools/rowln_bench.py [--abl CODE]"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--abl", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    shapes = [(a.batch * 197, 768), (a.batch * 77, 512)]
    P = []
    for M, N in shapes:
        P.append(dict(a=torch.randn(M, N, device=dev).to(torch.bfloat16), w=(torch.randn(N, N, device=dev) * N ** -0.5).to(torch.bfloat16),
                      bias=torch.randn(N, device=dev) * 0.1, x=torch.randn(M, N, device=dev), g=torch.ones(N, device=dev), b=torch.zeros(N, device=dev),
                      y=torch.empty(M, N, dtype=torch.bfloat16, device=dev)))
    # something between the launches that evicts the caches the way the step does: the MLP's 310 MB activation write
    for p in P:
        p["wk"] = ops.pack_w_ksteps(p["w"])
    scratch = torch.empty(a.batch * 197, 3072, dtype=torch.bfloat16, device=dev)

    def two():
        ops.gemm_bf16_grouped([(p["a"], p["w"], p["bias"], p["x"], p["x"]) for p in P], out_dtype=torch.float32)
        ops.add_layernorm_grouped([(p["x"], None, p["g"], p["b"], 1e-5, p["y"]) for p in P])

    def one():
        ops.gemm_residual_ln_grouped([(p["a"], p["wk"], p["bias"], p["x"], p["g"], p["b"], 1e-5, p["y"]) for p in P])

    def timed(fn):
        ts = []
        for i in range(a.iters + 5):
            scratch.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            if i >= 5:
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2], ts[0]

    out = {}
    for rnd in range(2):
        out[f"two_launches_us_{rnd}"] = timed(two)
        _lib.lib().mmamd_debug_set_gemm_knob(5, a.abl)
        out[f"one_launch_us_{rnd}"] = timed(one)
        _lib.lib().mmamd_debug_set_gemm_knob(5, 0)
    print(json.dumps({"what": "out-projection + residual + norm2, both towers, B = %d: (median, min) us" % a.batch, "abl": a.abl, **out}))


if __name__ == "__main__":
    main()
