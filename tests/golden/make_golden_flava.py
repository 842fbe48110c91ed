"""FLAVA fixtures from the REFERENCE (see make_golden.py for the mechanism):  python -m tests.golden.make_golden_flava
  flava_layer_kat.npz  reference TransformerEncoderLayer KAT (tests/models/flava/test_transformer.py:22-60): weights, input, output
  flava_small.npz      a kernel-legal small FLAVAModel (hidden 128, 2 heads of 64, 2/2/2 layers, 32x32 images, 16x16 patches,
                       16-token text with padding) with weights: every field of FLAVAOutput that the dual-encoder / mm path
                       produces + the global contrastive loss (with a row mask)
  flava_full_b2.npz    flava_model() (241 M parameters, seed 0) on a seeded batch of 2: projected embeddings, CLS rows, pooler
                       outputs, attention-prob and hidden-state checksums, key list + per-tensor checksums of the weights
"""
from __future__ import annotations

import random
import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden import _ref_shim  # noqa: E402
from tests.golden.make_golden import checksums, sd_np, seed  # noqa: E402

OUT = Path(__file__).resolve().parent
warnings.filterwarnings("ignore")


def tnp(t):
    return t.detach().numpy().copy()


def pack_output(prefix, o, store):
    store[prefix + "last_hidden_state"] = tnp(o.last_hidden_state)
    store[prefix + "pooler_output"] = tnp(o.pooler_output)
    store[prefix + "hidden_states"] = np.stack([tnp(h) for h in o.hidden_states])
    store[prefix + "attentions"] = np.stack([tnp(a) for a in o.attentions])


def main():
    _ref_shim.install()
    from torchmultimodal.models.flava.model import flava_model
    from torchmultimodal.models.flava.transformer import TransformerEncoderLayer
    from torchmultimodal.modules.losses.flava import FLAVAGlobalContrastiveLoss

    torch.set_num_threads(8)
    # ---- layer KAT (reference test_transformer.py: seed 4, layer(2,1,2), input randn(1,2,2,2,2))
    seed(4)
    x = torch.randn(1, 2, 2, 2, 2)
    layer = TransformerEncoderLayer(2, 1, 2, norm_first=True).eval()
    with torch.no_grad():
        y = layer(x)
    assert abs(float(y[0, 0, 0, 0, 0]) - (-1.5605)) < 1e-4
    np.savez(OUT / "flava_layer_kat.npz", x=tnp(x), y=tnp(y), **{"sd." + k: v for k, v in sd_np(layer).items()})

    # ---- small FLAVA model with weights
    seed(21)
    kw = dict(image_hidden_size=128, image_num_attention_heads=2, image_num_hidden_layers=2, image_intermediate_size=256,
              image_size=32, patch_size=16, text_hidden_size=128, text_num_attention_heads=2, text_num_hidden_layers=2,
              text_intermediate_size=256, vocab_size=200, max_position_embeddings=32, multimodal_hidden_size=128,
              multimodal_num_attention_heads=2, multimodal_num_hidden_layers=2, multimodal_intermediate_size=256,
              text_and_image_proj_size=64)
    model = flava_model(**kw).eval()
    g = torch.Generator().manual_seed(5)
    B = 5
    image = torch.randn(B, 3, 32, 32, generator=g)
    text = torch.randint(1, 200, (B, 16), generator=g)
    text[0, 10:] = 0
    text[3, 5:] = 0  # padding -> key mask
    text_masked = text.clone()
    text_masked[:, 2:4] = 103
    patches_mask = torch.randint(0, 2, (B, 4), generator=g)
    with torch.no_grad():
        out = model(image, text, image_patches_mask=patches_mask, text_masked=text_masked, skip_unmasked_mm_encoder=True)
        loss_mod = FLAVAGlobalContrastiveLoss().eval()
        mask = torch.tensor([True, True, False, True, True])
        lo = loss_mod(out.projected_image_embeddings, out.projected_text_embeddings, mask)
    st = {"image": tnp(image), "text": tnp(text), "text_masked": tnp(text_masked), "patches_mask": tnp(patches_mask),
          "proj_image": tnp(out.projected_image_embeddings), "proj_text": tnp(out.projected_text_embeddings), "loss_mask": tnp(mask),
          "itc_loss": tnp(lo.loss), "itc_image_logits": tnp(lo.image_logits), "itc_text_logits": tnp(lo.text_logits),
          "itc_image_embedding": tnp(lo.image_embedding), "itc_text_embedding": tnp(lo.text_embedding)}
    for name in ("image", "text", "image_masked", "text_masked", "multimodal_masked"):
        pack_output(name + ".", getattr(out, name), st)
    st.update({"sd." + k: v for k, v in sd_np(model).items()})
    np.savez_compressed(OUT / "flava_small.npz", **st)

    # ---- full-size flava_model(), seed 0
    seed(0)
    model = flava_model().eval()
    g = torch.Generator().manual_seed(77)
    image = torch.randn(2, 3, 224, 224, generator=g)
    text = torch.randint(1, 30500, (2, 77), generator=g)
    text[1, 40:] = 0
    with torch.no_grad():
        img, pi = model.encode_image(image, projection=True)
        txt, pt = model.encode_text(text, projection=True)
    k, s, a = checksums(model)
    np.savez(OUT / "flava_full_b2.npz", proj_image=tnp(pi), proj_text=tnp(pt), image_cls=tnp(img.last_hidden_state[:, 0]),
             text_cls=tnp(txt.last_hidden_state[:, 0]), image_pooler=tnp(img.pooler_output), text_pooler=tnp(txt.pooler_output),
             image_attn_last_sum=float(img.attentions[-1].double().sum()), text_attn_row=tnp(txt.attentions[-1][1, 0, 0]),
             image_hidden_last_mean=float(img.hidden_states[-1].double().mean()), keys=k, sums=s, asums=a,
             image_sum=float(image.double().sum()), text_sum=int(text.sum()))
    print("flava fixtures written")


if __name__ == "__main__":
    main()
