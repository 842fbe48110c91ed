"""Raw inputs -> loss: the headline forward (CLIP ViT-B/16 + contrastive loss, B = 256) with the input side inside the timed loop.

  resident   bench.py's regime: the fp32 image batch and the token ids already in HBM
  device     decoded uint8 500x375 images in HBM (a GPU JPEG decoder's output) + caption strings on the host:
             CLIPImageTransform.patches -> CLIPViTEncoder.forward_patches, CLIPTextTransform -> text tower, loss
  host       decoded uint8 images as numpy arrays on the host (PCIe + staging inclusive)
  host+pre   the same with the next batch's transform issued on a side stream while the current batch's towers run

    python tools/e2e_bench.py [--batch 256] [--steps 10]
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
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    from multimodal_amd import ops
    from multimodal_amd.models.clip import model as M
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.transforms.clip_transform import CLIPImageTransform, CLIPTextTransform
    from multimodal_amd.utils.synthetic import clip_batch

    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    model = M.clip_vit_b16().to(dev).eval()
    loss_fn = ContrastiveLossWithTemperature().to(dev)
    rng = np.random.default_rng(0)
    base = [rng.integers(0, 256, (375, 500, 3), dtype=np.uint8) for _ in range(16)]
    host = [base[i % 16] for i in range(a.batch)]
    devimgs = [torch.from_numpy(x).to(dev) for x in host]
    words = ["cat", "dog", "bicycle on a street in the rain", "very large aeroplane over the mountains at dusk"]
    texts = [f"a photo of a {words[i % 4]}, number {i}" for i in range(a.batch)]
    it = CLIPImageTransform(is_train=False)
    tt = CLIPTextTransform(text_bpe_merges_path=os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "clip_bpe_merges.txt.gz"),
                           device=dev)
    images, ids = clip_batch(a.batch)
    images, ids = images.to(dev), ids.to(dev)

    def towers(patches, token_ids):
        out = model.forward_patches(patches, token_ids)  # the same two-stream schedule as model(images, ids)
        return loss_fn(out.embeddings_a, out.embeddings_b)

    def resident():
        out = model(images, ids)
        return loss_fn(out.embeddings_a, out.embeddings_b)

    def from_device():
        return towers(it.patches(devimgs, 16, 768), tt(texts))

    def from_host():
        return towers(it.patches(host, 16, 768), tt(texts))

    side = torch.cuda.Stream()
    pending = {}

    def prefetch():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pending["p"], pending["t"] = it.patches(host, 16, 768), tt(texts)

    def from_host_prefetched():
        torch.cuda.current_stream().wait_stream(side)
        p, t = pending["p"], pending["t"]
        p.record_stream(torch.cuda.current_stream())
        loss = towers(p, t)
        prefetch()  # the next batch's input side overlaps this batch's towers
        return loss

    def timed(fn, setup=None):
        with torch.no_grad():
            if setup:
                setup()
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                loss = fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3, float(loss)

    out = {"workload": f"CLIP ViT-B/16 fwd + contrastive loss, B={a.batch}, raw 500x375 RGB images + caption strings -> loss"}
    for name, fn, setup in (("resident", resident, None), ("device", from_device, None), ("host", from_host, None),
                            ("host_prefetch", from_host_prefetched, prefetch)):
        ms, loss = timed(fn, setup)
        out[name + "_ms"] = round(ms, 3)
        out[name + "_pairs_per_s"] = round(a.batch / ms * 1e3, 1)
    out["loss"] = loss
    print(json.dumps(out))


if __name__ == "__main__":
    main()
