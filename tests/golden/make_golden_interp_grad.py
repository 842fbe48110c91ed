"""Gradients through `interpolate_pos_encoding` from the REFERENCE (models/flava/image_encoder.py:102-137,170-173):
python -m tests.golden.make_golden_interp_grad  ->  interp_grad.npz
  ImageEmbeddings(image_size=64, patch_size=16, hidden_size=128) — a 4 x 4 position grid — in TRAIN mode (dropout 0) on 96 x 96 images (6 x 6 patches,
  bicubic resampling of the trained table) with a patch mask: loss = sum(out * w) -> the gradient of position_embeddings, cls_token, mask_token and the
  patch projection (torch autograd through nn.functional.interpolate(mode="bicubic"))."""
from __future__ import annotations

import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden import _ref_shim  # noqa: E402
from tests.golden.make_golden import sd_np, seed  # noqa: E402

OUT = Path(__file__).resolve().parent
warnings.filterwarnings("ignore")


def tnp(t):
    return t.detach().numpy().copy()


def main():
    _ref_shim.install()
    from torchmultimodal.models.flava.image_encoder import ImageEmbeddings

    torch.set_num_threads(8)
    seed(71)
    emb = ImageEmbeddings(image_size=64, patch_size=16, hidden_size=128, use_image_masking=True).train()
    with torch.no_grad():  # the reference initialises these three to zeros: give them values so their gradients mean something
        emb.position_embeddings.normal_(std=0.5)
        emb.cls_token.normal_(std=0.5)
        emb.mask_token.normal_(std=0.5)
    g = torch.Generator().manual_seed(5)
    image = torch.randn(3, 3, 96, 96, generator=g)
    pm = (torch.rand(3, 36, generator=g) < 0.3).long()
    w = torch.randn(3, 37, 128, generator=g)
    out = emb(image, image_patches_mask=pm, interpolate_pos_encoding=True)
    (out * w).sum().backward()
    st = {"image": tnp(image), "patches_mask": tnp(pm), "w": tnp(w), "out": tnp(out)}
    st.update({"sd." + k: v for k, v in sd_np(emb).items()})
    st.update({"g." + k: tnp(p.grad) for k, p in emb.named_parameters()})
    np.savez_compressed(OUT / "interp_grad.npz", **st)
    print("interp_grad.npz", out.shape, sorted(k for k in st if k.startswith("g.")))


if __name__ == "__main__":
    main()
