"""Input-side micro-benchmark: CLIPImageTransform on B decoded 500x375 RGB images -> bf16 patch rows.

  device   the decoded uint8 images already in HBM (e.g. from a GPU JPEG decoder): the two resampling kernels + descriptor copy
  host     numpy images on the host: + pinned staging and the H2D copy (PCIe-inclusive)
  cpu      the reference's per-image host path restated with Pillow + numpy on ONE core (Resize, CenterCrop, ToTensor, Normalize)
           on a bounded sample

    python tools/transform_bench.py [--batch 256] [--iters 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cpu-sample", type=int, default=64)
    a = ap.parse_args()
    from multimodal_amd.transforms.clip_transform import CLIP_DEFAULT_MEAN, CLIP_DEFAULT_STD, CLIPImageTransform, CLIPTextTransform

    rng = np.random.default_rng(0)
    base = [rng.integers(0, 256, (375, 500, 3), dtype=np.uint8) for _ in range(16)]
    host = [base[i % 16] for i in range(a.batch)]
    dev = [torch.from_numpy(x).cuda() for x in host]
    t = CLIPImageTransform(is_train=False)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.iters * 1e3

    ms_dev = timed(lambda: t.patches(dev, 16, 768))
    ms_host = timed(lambda: t.patches(host, 16, 768))
    src_bytes = a.batch * 375 * 500 * 3
    out = {"workload": f"CLIPImageTransform(eval) {a.batch} x 500x375 RGB -> bf16 patch rows [B*196,768]",
           "device_ms": round(ms_dev, 3), "device_images_per_s": round(a.batch / ms_dev * 1e3, 1),
           "device_source_GBps": round(src_bytes / ms_dev / 1e6, 1),
           "host_ms": round(ms_host, 3), "host_images_per_s": round(a.batch / ms_host * 1e3, 1)}
    try:
        from PIL import Image

        pil = [Image.fromarray(x) for x in host[: a.cpu_sample]]
        mean = np.asarray(CLIP_DEFAULT_MEAN, np.float32)[:, None, None]
        std = np.asarray(CLIP_DEFAULT_STD, np.float32)[:, None, None]
        t0 = time.perf_counter()
        for p in pil:
            r = p.resize((298, 224), Image.BICUBIC).crop((37, 0, 261, 224)).convert("RGB")
            x = np.asarray(r).astype(np.float32).transpose(2, 0, 1) / np.float32(255)
            x = (x - mean) / std
        dt = time.perf_counter() - t0
        out["cpu_images_per_s_1core"] = round(len(pil) / dt, 1)
        out["cpu_sample"] = f"{len(pil)} images, Pillow {Image.__version__ if hasattr(Image, '__version__') else ''} + numpy, 1 thread"
    except ImportError:
        out["cpu_images_per_s_1core"] = None
    texts = ["a photo of a " + w for w in ("cat", "dog", "bicycle on a street in the rain", "very large aeroplane")] * (a.batch // 4)
    tt = CLIPTextTransform(text_bpe_merges_path=os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "clip_bpe_merges.txt.gz"),
                           device="cuda")
    tt(texts)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        ids = tt(texts)
    torch.cuda.synchronize()
    out["text_ms_per_batch"] = round((time.perf_counter() - t0) / a.iters * 1e3, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
