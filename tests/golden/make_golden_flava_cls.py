"""FLAVAForClassification + interpolate_pos_encoding fixtures from the REFERENCE:  python -m tests.golden.make_golden_flava_cls
  flava_cls_interp.npz  small flava_model_for_classification (weights, inputs): logits + loss for required_embedding image / text / mm,
                        gradients of the classifier and of a few encoder tensors (classifier_dropout = 0, train mode);
                        ImageEmbeddings(..., interpolate_pos_encoding=True) on 48x48 images; the full-size [1,197,768] table
                        interpolated to 160x160 and 96x96 inputs
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

KW = dict(image_hidden_size=128, image_num_attention_heads=2, image_num_hidden_layers=2, image_intermediate_size=256,
          image_size=32, patch_size=16, text_hidden_size=128, text_num_attention_heads=2, text_num_hidden_layers=2,
          text_intermediate_size=256, vocab_size=200, max_position_embeddings=32, multimodal_hidden_size=128,
          multimodal_num_attention_heads=2, multimodal_num_hidden_layers=2, multimodal_intermediate_size=256,
          text_and_image_proj_size=64)
GRAD_KEYS = ["classifier.model.0.weight", "classifier.model.0.bias", "classifier.model.2.weight", "classifier.model.2.bias",
             "model.image_encoder.layernorm.weight", "model.image_encoder.embeddings.cls_token",
             "model.image_encoder.encoder.layer.1.feedforward.model.0.weight"]


def tnp(t):
    return t.detach().numpy().copy()


def main():
    _ref_shim.install()
    from torchmultimodal.models.flava.image_encoder import ImageEmbeddings
    from torchmultimodal.models.flava.model import flava_model_for_classification

    torch.set_num_threads(8)
    seed(31)
    model = flava_model_for_classification(num_classes=7, classifier_in_dim=128, classifier_hidden_sizes=16, classifier_dropout=0.0,
                                           pretrained=False, **KW)
    g = torch.Generator().manual_seed(8)
    B = 6
    image = torch.randn(B, 3, 32, 32, generator=g)
    text = torch.randint(1, 200, (B, 16), generator=g)
    text[1, 9:] = 0
    labels = torch.randint(0, 7, (B,), generator=g)
    # the gradient comparison is only meaningful if no hidden unit sits on the ReLU kink: a bf16-level difference in the CLS row
    # would flip its mask.  Re-draw the classifier until every pre-activation of the image-mode step is at least 0.03 from zero.
    with torch.no_grad():
        cls_row = model.eval().model(image=image, required_embedding="image", skip_unmasked_mm_encoder=False).image.last_hidden_state[:, 0]
        for trial in range(2000):
            torch.manual_seed(1000 + trial)
            for m in model.classifier.model:
                if hasattr(m, "reset_parameters"):
                    m.reset_parameters()
            pre = model.classifier.model[0](cls_row)
            if float(pre.abs().min()) > 0.03:
                break
        else:
            raise SystemExit("no classifier draw with a safe ReLU margin")
        print("classifier draw", trial, "min |pre-activation|", float(pre.abs().min()))
    st = {"image": tnp(image), "text": tnp(text), "labels": tnp(labels)}
    st.update({"sd." + k: v for k, v in sd_np(model).items()})
    model.eval()
    with torch.no_grad():
        for mode, kw in (("image", dict(image=image)), ("text", dict(text=text)), ("mm", dict(image=image, text=text))):
            o = model(required_embedding=mode, labels=labels, **kw)
            st[f"{mode}.logits"], st[f"{mode}.loss"] = tnp(o.logits), tnp(o.loss)
        o = model(image=image, text=text, required_embedding="mm", labels=labels, cls_index=3)
        st["mm.cls3.logits"] = tnp(o.logits)
    model.train()
    model.zero_grad()
    o = model(image=image, required_embedding="image", labels=labels)
    o.loss.backward()
    named = dict(model.named_parameters())
    for k in GRAD_KEYS:
        st["grad." + k] = tnp(named[k].grad)
    st["train.image.loss"] = tnp(o.loss)

    # interpolation on the small model's embeddings (2x2 grid -> 3x3)
    emb = model.model.image_encoder.embeddings.eval()
    big = torch.randn(2, 3, 48, 48, generator=g)
    with torch.no_grad():
        st["interp.image48"] = tnp(big)
        st["interp.emb48"] = tnp(emb(big, interpolate_pos_encoding=True))
    # full-size table
    seed(5)
    full = ImageEmbeddings(image_size=224, patch_size=16, hidden_size=768)
    with torch.no_grad():
        full.position_embeddings.normal_(std=0.02)
    st["interp.full_pos"] = tnp(full.position_embeddings)
    for side in (160, 96):
        n = (side // 16) ** 2
        with torch.no_grad():
            st[f"interp.full_{side}"] = tnp(full.interpolate_pos_encoding(torch.zeros(1, n + 1, 768), side, side))
    np.savez_compressed(OUT / "flava_cls_interp.npz", **st)
    print({k: v.shape for k, v in st.items() if not k.startswith("sd.")})
    print("bytes", (OUT / "flava_cls_interp.npz").stat().st_size)


if __name__ == "__main__":
    main()
