"""FLAVA image codebook (DALL-E dVAE encoder, 8192 codes, 112x112 inputs) timing on one MI355X:  python tools/codebook_bench.py [--batch 128]"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def encoder_gflop(vae, H=112, W=112):
    """2 * MACs of every convolution of the encoder for one image."""
    from multimodal_amd.models.flava.model import DalleConv2d

    enc = vae.encoder
    fl, h, w = 0.0, H, W
    for name, mod in enc.blocks.named_children():
        for m in mod.modules():
            if isinstance(m, DalleConv2d):
                fl += 2.0 * h * w * m.w.numel()
        if any(isinstance(m, torch.nn.MaxPool2d) for m in mod.modules()):
            h, w = h // 2, w // 2
    return fl / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    from multimodal_amd.models.flava.model import DalleVAEEncoder

    torch.manual_seed(0)
    vae = DalleVAEEncoder(pretrained=False).cuda().eval()
    x = torch.randn(a.batch, 3, 112, 112, device="cuda")
    with torch.no_grad():
        for _ in range(a.warmup):
            ids = vae(x)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(a.steps):
            ids = vae(x)
        t1.record()
        torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / a.steps
    gf = encoder_gflop(vae)
    print(json.dumps({"workload": "DalleVAEEncoder.get_codebook_indices, 112x112 -> 14x14 codes of 8192 (random weights)", "batch": a.batch,
                      "ms_per_step": round(ms, 3), "images_per_s": round(a.batch / ms * 1e3, 1), "gflop_per_image": round(gf, 2),
                      "tflops": round(a.batch * gf / ms, 1), "mfma_frac": round(a.batch * gf / ms / 2500.0, 4),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "codes_used": int(ids.unique().numel())}))


if __name__ == "__main__":
    main()
