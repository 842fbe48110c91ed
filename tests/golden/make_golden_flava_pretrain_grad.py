"""Full FLAVA pre-training step gradients from the REFERENCE (torch autograd):  python -m tests.golden.make_golden_flava_pretrain_grad
  flava_pretrain_grad.npz  small FLAVAModel (weights of flava_small.npz) + FLAVAPretrainingLoss (weights of flava_pretrain_small.npz),
                           train mode, image / text / text_masked / patches_mask of flava_small.npz and the labels of
                           flava_pretrain_small.npz: total loss = sum of ITM + MMM-text + MMM-image + global contrastive, backward:
                           the four loss values and the gradient of every parameter of the model and of the loss module.
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
from tests.golden.make_golden_flava_grad import SMALL_KW  # noqa: E402

OUT = Path(__file__).resolve().parent
warnings.filterwarnings("ignore")


def main():
    _ref_shim.install()
    from torchmultimodal.models.flava.model import flava_model
    from torchmultimodal.modules.losses.flava import FLAVAPretrainingLoss

    z, zl = np.load(OUT / "flava_small.npz"), np.load(OUT / "flava_pretrain_small.npz")
    model = flava_model(**SMALL_KW)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}, strict=True)
    loss = FLAVAPretrainingLoss(hidden_size=128, text_vocab_size=200, image_vocab_size=64)
    loss.load_state_dict({k[3:]: torch.from_numpy(zl[k]) for k in zl.files if k.startswith("sd.")}, strict=True)
    model.train(); loss.train()
    T = lambda a: torch.from_numpy(np.asarray(a))
    out = model(T(z["image"]), T(z["text"]), image_patches_mask=T(z["patches_mask"]), text_masked=T(z["text_masked"]))
    lo = loss(image_sequence=out.image.last_hidden_state, text_sequence=out.text.last_hidden_state,
              image_masked_sequence=out.image_masked.last_hidden_state, text_masked_sequence=out.text_masked.last_hidden_state,
              multimodal_masked_sequence=out.multimodal_masked.last_hidden_state, itm_labels=T(zl["itm_labels"]),
              mim_labels=T(zl["mim_labels"]), mlm_labels=T(zl["mlm_labels"]),
              projected_image_embeddings=out.projected_image_embeddings, projected_text_embeddings=out.projected_text_embeddings)
    names = ("itm_loss", "mmm_text_loss", "mmm_image_loss", "global_contrastive_loss")
    total = sum(getattr(lo.losses, n) for n in names)
    total.backward()
    st = {n: getattr(lo.losses, n).detach().numpy() for n in names}
    none = []
    for prefix, mod in (("model.", model), ("loss.", loss)):
        for k, p in mod.named_parameters():
            if p.grad is None:
                none.append(prefix + k)
            else:
                st["g." + prefix + k] = p.grad.numpy()
    st["no_grad_keys"] = np.array(none)
    np.savez_compressed(OUT / "flava_pretrain_grad.npz", **st)
    print("written", len(st), "arrays;", {n: float(st[n]) for n in names}, "no grad:", len(none))


if __name__ == "__main__":
    main()
