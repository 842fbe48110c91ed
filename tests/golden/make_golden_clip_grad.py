"""CLIP training-step gradients from the REFERENCE (torch autograd):  python -m tests.golden.make_golden_clip_grad
  clip_grad.npz  the midsize two-tower model of midsize.npz (same weights, same batch), train mode, loss =
                 ContrastiveLossWithTemperature(CLIP(images, ids)), loss.backward(): the loss, every parameter's gradient
                 (126 tensors incl. logit_scale) and the gradients w.r.t. the two embedding outputs.
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


def main():
    _ref_shim.install()
    from torchmultimodal.models.clip.image_encoder import CLIPViTEncoder
    from torchmultimodal.models.clip.model import CLIP
    from torchmultimodal.models.clip.text_encoder import CLIPTextEncoder
    from torchmultimodal.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature

    z = np.load(OUT / "midsize.npz")
    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=2, patch_size=16, image_size=64, width=128)
    txt = CLIPTextEncoder(embedding_dim=64, context_length=77, vocab_size=1000, width=128, dim_feedforward=256, heads=2, layers=2)
    clip = CLIP(vit, txt)
    clip.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}, strict=True)
    clip.train()
    loss_fn = ContrastiveLossWithTemperature()
    out = clip(torch.from_numpy(z["images"]), torch.from_numpy(z["ids"]))
    out.embeddings_a.retain_grad(); out.embeddings_b.retain_grad()
    loss = loss_fn(out.embeddings_a, out.embeddings_b)
    loss.backward()
    st = {"loss": loss.detach().numpy(), "grad_emb_a": out.embeddings_a.grad.numpy(), "grad_emb_b": out.embeddings_b.grad.numpy(),
          "g.logit_scale": loss_fn.logit_scale.grad.numpy()}
    for k, p in clip.named_parameters():
        assert p.grad is not None, k
        st["g." + k] = p.grad.numpy()
    assert abs(float(loss) - float(z["loss"])) < 1e-4  # dropout is 0 in these towers: train == eval forward
    np.savez_compressed(OUT / "clip_grad.npz", **st)
    print("written", len(st), "arrays; loss", float(loss))


if __name__ == "__main__":
    main()
