"""Same-process alternating A/B of the LayerNorm input-load policy (mmamd_debug_set_attn_variant(3100 + p): 1 = never non-temporal, 2 = always) on the other configurations:
FLAVA B = 128 forward, CLIP ViT-L/14 B = 256, ViT-B/32 B = 256, CoCa L/14 B = 128 forward + losses.   python tools/nt_ab_models.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")


def timed(step, n):
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def ab(name, step, mask, n=8, rounds=3):
    res = {1: [], mask: []}
    for _ in range(rounds):
        for m in (1, mask):
            L.mmamd_debug_set_attn_variant(3100 + m)
            res[m].append(timed(step, n))
    L.mmamd_debug_set_attn_variant(3100)
    med = {m: sorted(v)[len(v) // 2] for m, v in res.items()}
    print(f"{name:24s} cached loads: {med[1]:8.3f} ms   non-temporal loads: {med[mask]:8.3f} ms   ({(med[mask] / med[1] - 1) * 100:+.2f} %)", flush=True)


@torch.no_grad()
def main():
    mask = 2
    g = torch.Generator().manual_seed(1)
    from multimodal_amd.models.flava.model import flava_model
    torch.manual_seed(0)
    model = flava_model().to(dev).eval()
    B = 128
    image = torch.randn(B, 3, 224, 224, generator=g).to(dev)
    text = torch.randint(1, 30522, (B, 77), generator=g)
    text[:, 60:] = 0
    tm = text.clone()
    tm[:, 5:12] = 103
    pm = (torch.rand(B, 14, 14, generator=g) < 0.4).to(dev)
    text, tm = text.to(dev), tm.to(dev)
    ab("flava B=128 fwd", lambda: model(image, text, image_patches_mask=pm, text_masked=tm, skip_unmasked_mm_encoder=True), mask)
    del model
    from multimodal_amd.models.clip import clip_vit_b32, clip_vit_l14
    from multimodal_amd.utils.synthetic import clip_batch
    images, ids = clip_batch(256)
    images, ids = images.to(dev).to(torch.bfloat16), ids.to(dev)
    for nm, ctor in (("clip l14 B=256", clip_vit_l14), ("clip b32 B=256", clip_vit_b32)):
        torch.manual_seed(0)
        m = ctor().to(dev).eval()
        ab(nm, lambda: m(images, ids), mask)
        del m
    from tools.coca_bench import L14
    from multimodal_amd.models.coca.coca_model import CoCaForPretraining, coca_vit
    torch.manual_seed(0)
    cm = CoCaForPretraining(coca_vit(**L14)).to(dev).eval()
    ci = torch.randn(128, 3, 224, 224, generator=g).to(dev)
    ct = torch.randint(1, 49408, (128, 77), generator=g)
    ct[:, 60:] = 0
    ct = ct.to(dev)
    ab("coca l14 B=128 fwd+loss", lambda: cm(ci, ct), mask, n=5)


if __name__ == "__main__":
    main()
