"""Yardstick only (never on the product path): what the vendor BLAS behind torch.matmul (hipBLASLt / rocBLAS) reaches on the four ViT-B/16 B = 256
projection shapes and the text tower's, next to mmamd_gemm_bf16 on the same operands, same box, interleaved.  Plain GEMM without bias / activation /
residual for the library (its epilogues differ), the shipped epilogue for ours.
    python tools/blas_yardstick.py [--rounds 5]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402

SHAPES = [("vit qkv", 50432, 2304, 768, "bf16"), ("vit out-proj", 50432, 768, 768, "f32res"), ("vit MLP-up", 50432, 3072, 768, "gelu"),
          ("vit MLP-down", 50432, 768, 3072, "f32res"), ("text qkv", 19712, 1536, 512, "bf16"), ("text MLP-up", 19712, 2048, 512, "gelu")]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = ops.StreamTimer()
    t.start()
    for _ in range(reps):
        fn()
    t.stop()
    return t.elapsed_ms() / reps * 1e3


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda")
    print(f"{'shape':14s} {'M':>6s} {'N':>5s} {'K':>5s} | torch.matmul us (TF/s) | mmamd plain us (TF/s) | mmamd shipped epilogue us (TF/s)")
    with torch.no_grad():
        for name, M, N, K, kind in SHAPES:
            x = torch.randn(M, K, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
            b = torch.randn(N, device=dev)
            wt = w.t()
            o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            r = torch.randn(M, N, device=dev) if kind == "f32res" else None
            best = [1e9, 1e9, 1e9]
            for _ in range(a.rounds):
                best[0] = min(best[0], timed(lambda: torch.matmul(x, wt, out=o)))
                best[1] = min(best[1], timed(lambda: ops.gemm_bf16(x, w, None, out=o)))
                if kind == "f32res":
                    best[2] = min(best[2], timed(lambda: ops.gemm_bf16(x, w, b, residual=r, out=r, out_dtype=torch.float32)))
                else:
                    best[2] = min(best[2], timed(lambda: ops.gemm_bf16(x, w, b, act=ops.ACT_QUICKGELU if kind == "gelu" else ops.ACT_NONE, out=o)))
            fl = 2.0 * M * N * K
            print(f"{name:14s} {M:6d} {N:5d} {K:5d} | " + " | ".join(f"{t:7.1f} ({fl / t / 1e6:6.0f})" for t in best), flush=True)
