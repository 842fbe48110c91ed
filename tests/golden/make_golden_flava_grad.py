"""FLAVA training-step gradients from the REFERENCE (torch autograd):  python -m tests.golden.make_golden_flava_grad
  flava_grad.npz  the small FLAVAModel of flava_small.npz (same weights, image / text / text_masked of that fixture, no patch mask),
                  train mode: L = FLAVAGlobalContrastiveLoss(projected image, projected text, all rows) + mean_b(mm_masked CLS row . linspace(-1,1,d));
                  L.backward(): the two loss terms and the gradient of every parameter that is reached (poolers are not).
"""
from __future__ import annotations

import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden import _ref_shim  # noqa: E402

OUT = Path(__file__).resolve().parent
warnings.filterwarnings("ignore")

SMALL_KW = dict(image_hidden_size=128, image_num_attention_heads=2, image_num_hidden_layers=2, image_intermediate_size=256,
                image_size=32, patch_size=16, text_hidden_size=128, text_num_attention_heads=2, text_num_hidden_layers=2,
                text_intermediate_size=256, vocab_size=200, max_position_embeddings=32, multimodal_hidden_size=128,
                multimodal_num_attention_heads=2, multimodal_num_hidden_layers=2, multimodal_intermediate_size=256,
                text_and_image_proj_size=64)


def main():
    _ref_shim.install()
    from torchmultimodal.models.flava.model import flava_model
    from torchmultimodal.modules.losses.flava import FLAVAGlobalContrastiveLoss

    z = np.load(OUT / "flava_small.npz")
    model = flava_model(**SMALL_KW)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}, strict=True)
    model.train()
    loss_mod = FLAVAGlobalContrastiveLoss()
    image, text, text_masked = (torch.from_numpy(z[k]) for k in ("image", "text", "text_masked"))
    out = model(image, text, text_masked=text_masked)
    mask = torch.ones(image.shape[0], dtype=torch.bool)
    itc = loss_mod(out.projected_image_embeddings, out.projected_text_embeddings, mask).loss
    probe = torch.linspace(-1.0, 1.0, out.multimodal_masked.last_hidden_state.shape[-1])  # fixed direction: a non-trivial gradient through the final LayerNorm
    mm_term = (out.multimodal_masked.last_hidden_state[:, 0] * probe).sum(-1).mean()
    (itc + mm_term).backward()
    st = {"itc": itc.detach().numpy(), "mm_term": mm_term.detach().numpy(), "g.logit_scale": loss_mod.logit_scale.grad.numpy()}
    none = []
    for k, p in model.named_parameters():
        if p.grad is None:
            none.append(k)
        else:
            st["g." + k] = p.grad.numpy()
    st["no_grad_keys"] = np.array(none)
    np.savez_compressed(OUT / "flava_grad.npz", **st)
    print("written", len(st), "arrays; itc", float(itc), "mm", float(mm_term), "params without grad:", none)


if __name__ == "__main__":
    main()
