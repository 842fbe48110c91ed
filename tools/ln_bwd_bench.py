"""mmamd_layernorm_bwd alone on the two shapes of the CLIP ViT-B/16 training step (B = 256): bytes moved per call and the rate, for fp32 and bf16 dy, with and
without the residual-path gradient.    python tools/ln_bwd_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402


def main():
    for name, rows, d in (("ViT 50432 x 768", 50432, 768), ("text 19712 x 512", 19712, 512), ("L/14 65792 x 1024", 65792, 1024)):
        torch.manual_seed(0)
        x = torch.randn(rows, d, device="cuda")
        g = torch.randn(d, device="cuda")
        add = torch.randn(rows, d, device="cuda")
        for dy_dt in (torch.float32, torch.bfloat16):
            dy = torch.randn(rows, d, device="cuda").to(dy_dt)
            for with_add in (True, False):
                pend = []
                fn = lambda: (ops.layernorm_bwd(x, g, dy, 1e-5, add=add if with_add else None, want_bf16=True, want_colsum=True, defer=pend), pend.clear())  # noqa: E731
                us = timeit(fn, 50) * 1000.0
                mb = rows * d * (4 + dy.element_size() + (4 if with_add else 0) + 4 + 2) / 1e6
                print(f"{name:20s} dy {str(dy_dt)[6:]:8s} add={with_add!s:5s}  {us:7.1f} us  {mb:6.0f} MB  {mb / us:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
