"""FLAVA pre-training loss fixture from the REFERENCE:  python -m tests.golden.make_golden_flava_loss
  flava_pretrain_small.npz  FLAVAPretrainingLoss(hidden 128, text vocab 200, image vocab 64) with weights, applied to the
                            sequences the reference FLAVAModel produced for flava_small.npz: the multimodal case (ITM + MMM
                            text/image + global contrastive with the ITM row filter) and the unimodal case (MIM + MLM)
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
from tests.golden.make_golden import sd_np, seed  # noqa: E402

OUT = Path(__file__).resolve().parent
warnings.filterwarnings("ignore")


def tnp(t):
    return t.detach().numpy().copy()


def main():
    _ref_shim.install()
    from torchmultimodal.modules.losses.flava import FLAVAPretrainingLoss

    z = np.load(OUT / "flava_small.npz")
    T = lambda k: torch.from_numpy(z[k])
    seed(33)
    loss = FLAVAPretrainingLoss(hidden_size=128, text_vocab_size=200, image_vocab_size=64).eval()
    for p in loss.parameters():  # non-trivial biases / LN affine so that every term is exercised
        if p.dim() == 1:
            p.data.add_(torch.randn_like(p) * 0.05)
    g = torch.Generator().manual_seed(8)
    B = z["text"].shape[0]
    mlm = torch.full((B, 16), -1, dtype=torch.long)
    mlm[:, 2:4] = torch.randint(1, 200, (B, 2), generator=g)
    mlm[1, 9] = 17
    mim = torch.randint(0, 64, (B, 4), generator=g)
    mim[T("patches_mask") == 0] = -1
    itm = torch.tensor([1, 0, 1, 1, 0])
    st = {"mlm_labels": tnp(mlm), "mim_labels": tnp(mim), "itm_labels": tnp(itm)}
    with torch.no_grad():
        mm = loss(image_sequence=T("image.last_hidden_state"), text_sequence=T("text.last_hidden_state"),
                  image_masked_sequence=T("image_masked.last_hidden_state"), text_masked_sequence=T("text_masked.last_hidden_state"),
                  multimodal_masked_sequence=T("multimodal_masked.last_hidden_state"), itm_labels=itm, mim_labels=mim, mlm_labels=mlm,
                  projected_image_embeddings=T("proj_image"), projected_text_embeddings=T("proj_text"))
        uni = loss(image_masked_sequence=T("image_masked.last_hidden_state"), text_masked_sequence=T("text_masked.last_hidden_state"),
                   mim_labels=mim, mlm_labels=mlm)
        allneg = loss(multimodal_masked_sequence=T("multimodal_masked.last_hidden_state"), itm_labels=torch.zeros(B, dtype=torch.long),
                      mim_labels=mim, mlm_labels=mlm, image_masked_sequence=T("image_masked.last_hidden_state"),
                      text_masked_sequence=T("text_masked.last_hidden_state"))
    for name in ("mmm_text_loss", "mmm_image_loss", "itm_loss", "global_contrastive_loss"):
        st["mm." + name] = tnp(getattr(mm.losses, name))
    assert mm.losses.mim_loss is None and mm.losses.mlm_loss is None
    st["mm.mmm_text_logits"] = tnp(mm.mmm_text_output.logits)
    st["mm.mmm_image_logits"] = tnp(mm.mmm_image_output.logits)
    st["mm.itm_logits"] = tnp(mm.itm_output.logits)
    st["mm.itc_image_logits"] = tnp(mm.global_contrastive_output.image_logits)
    st["uni.mim_loss"], st["uni.mlm_loss"] = tnp(uni.losses.mim_loss), tnp(uni.losses.mlm_loss)
    st["uni.mim_logits"], st["uni.mlm_logits"] = tnp(uni.mim_output.logits), tnp(uni.mlm_output.logits)
    st["allneg.itm_loss"], st["allneg.mmm_text_loss"] = tnp(allneg.losses.itm_loss), tnp(allneg.losses.mmm_text_loss)
    st["allneg.mmm_text_logits"] = tnp(allneg.mmm_text_output.logits)
    st.update({"sd." + k: v for k, v in sd_np(loss).items()})
    np.savez_compressed(OUT / "flava_pretrain_small.npz", **st)
    print("written", {k: v.shape for k, v in st.items() if not k.startswith("sd.")})


if __name__ == "__main__":
    main()
