"""CoCa fixtures from the REFERENCE (see make_golden.py for the mechanism):  python -m tests.golden.make_golden_coca
  coca_kat.npz        the reference's own constant-weight KAT model (tests/models/coca/test_coca_model.py:45-164): weights = 1,
                      expected 0.3536 / 8.0 and losses 0.6931 / 3.9120 — inputs, outputs, losses
  coca_small.npz      kernel-legal small coca_vit models with seeded random weights (re-created from the seed by the tests) (hidden 128 = 2 heads of 64; pooler with
                      2 heads of 64), parallel pooler and cascaded pooler, padded texts: MultimodalOutput + CoCaForPretraining losses
  coca_pool96.npz     same with a 192-wide pooler (2 heads of 96, the ViT-L/14 pooler head width) and projection to a vocabulary
  coca_l14_meta.npz   coca_vit_l_14() state_dict keys/shapes (no weights: 600+ M parameters) for the host-API test
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
from tests.golden.make_golden import checksums, sd_np, seed  # noqa: E402

OUT = Path(__file__).resolve().parent
warnings.filterwarnings("ignore")


def tnp(t):
    return t.detach().numpy().copy()


SMALL = dict(vision_patch_size=16, vision_dim_feedforward=256, vision_n_layer=2, vision_n_head=2, vocab_size=96, num_text_positions=13,
             text_hidden_dim=128, text_n_layer=2, text_n_head=2, text_dim_feedforward=256, text_output_dim=128, fusion_n_layer=2,
             fusion_n_head=2, fusion_dim_feedforward=256, multimodal_output_projection_dim=96, pooler_input_embed_dim=128,
             pooler_output_embed_dim=128, image_size=64, pooler_n_head=2, pooler_n_queries=6)
POOL96 = dict(SMALL, pooler_output_embed_dim=192, text_hidden_dim=192, text_n_head=3, text_output_dim=192, fusion_n_head=3,
              text_dim_feedforward=384, fusion_dim_feedforward=384)


def randomize(model, g):
    """coca_vit leaves several tensors at zero/constant init (position embeddings, biases): perturb everything so each term matters."""
    for p in model.parameters():
        p.data.add_(torch.randn(p.shape, generator=g) * (0.02 if p.dim() > 1 else 0.05))


def run(model_kw, cascaded, seed_v, store, prefix):
    from torchmultimodal.models.coca.coca_model import coca_vit, CoCaForPretraining

    seed(seed_v)
    model = coca_vit(**model_kw, cascaded_pooler=cascaded).eval()
    g = torch.Generator().manual_seed(seed_v + 1)
    randomize(model, g)
    B = 3
    S = model_kw["num_text_positions"]
    images = torch.randn(B, 3, model_kw["image_size"], model_kw["image_size"], generator=g)
    texts = torch.randint(1, model_kw["vocab_size"], (B, S), generator=g)
    texts[0, 7:] = 0
    texts[2, 10:] = 0
    with torch.no_grad():
        out = model(images, texts)
        store[prefix + "image_pooled_output"] = tnp(out.image_pooled_output)
        store[prefix + "text_pooled_output"] = tnp(out.text_pooled_output)
        store[prefix + "multimodal_embeddings"] = tnp(out.multimodal_embeddings)
        if not cascaded:  # the reference's loss matmul fails on the cascaded pooler's [B,1,D] output (SURVEY 8a note)
            pre = CoCaForPretraining(model).eval()
            losses = pre(images, texts)
            store[prefix + "loss_contrastive"], store[prefix + "loss_captioning"] = tnp(losses["contrastive"]), tnp(losses["captioning"])
            pm = texts != 0
            out_pm = model(images, texts, pm)
            assert torch.equal(out_pm.text_pooled_output, out.text_pooled_output)
    store[prefix + "images"], store[prefix + "texts"] = tnp(images), tnp(texts)
    # weights are NOT stored (5 MB per model): the tests rebuild them with the same seed + randomize() through the drop-in
    # modules (whose seeded initialisation is the reference's, tensor for tensor) and verify these per-tensor checksums
    k, sm, asm = checksums(model)
    store[prefix + "keys"], store[prefix + "sums"], store[prefix + "asums"] = k, sm, asm


def main():
    _ref_shim.install()

    def init_weights_with_constant(model, constant=1.0):  # what the reference's tests/test_utils.py:193-205 does
        for n, p in model.named_parameters():
            torch.nn.init.constant_(p, constant)
            if any(n.endswith(k) for k in ("text_projection.bias", "pooled_projection.bias", "output_projection.bias", "vision_proj.bias")):
                torch.nn.init.constant_(p, 0.0)

    from torchmultimodal.models.coca.coca_model import coca_vit, coca_vit_l_14, CoCaForPretraining

    torch.set_num_threads(8)
    # ---- the reference's KAT
    seed(0)
    kat = coca_vit(vision_patch_size=4, vision_dim_feedforward=24, vision_n_layer=2, vision_n_head=2, vocab_size=50, num_text_positions=11,
                   text_hidden_dim=8, text_n_layer=2, text_n_head=2, text_dim_feedforward=32, text_output_dim=8, fusion_n_layer=2,
                   fusion_n_head=2, fusion_dim_feedforward=32, multimodal_output_projection_dim=50, pooler_input_embed_dim=6,
                   pooler_output_embed_dim=8, image_size=12, pooler_n_head=2, cascaded_pooler=False)
    init_weights_with_constant(kat)
    kat.eval()
    texts = torch.LongTensor([[1, 3, 4, 5, 6, 7, 8, 2, 0, 0, 0], [1, 25, 28, 34, 39, 45, 40, 5, 12, 6, 2]])
    images = torch.randn(2, 3, 12, 12)
    with torch.no_grad():
        out = kat(images, texts)
        pre = CoCaForPretraining(kat)
        init_weights_with_constant(pre)
        losses = pre.eval()(images, texts)
    assert abs(float(out.image_pooled_output[0, 0]) - 0.3536) < 1e-4 and abs(float(out.multimodal_embeddings[0, 0, 0]) - 8.0) < 1e-4
    assert abs(float(losses["contrastive"]) - 0.6931) < 1e-4 and abs(float(losses["captioning"]) - 3.9120) < 1e-4
    st = {"images": tnp(images), "texts": tnp(texts), "image_pooled_output": tnp(out.image_pooled_output),
          "text_pooled_output": tnp(out.text_pooled_output), "multimodal_embeddings": tnp(out.multimodal_embeddings),
          "loss_contrastive": tnp(losses["contrastive"]), "loss_captioning": tnp(losses["captioning"]),
          "logit_scale": tnp(pre.contrastive_loss.logit_scale)}
    st.update({"sd." + k: v for k, v in sd_np(kat).items()})
    np.savez_compressed(OUT / "coca_kat.npz", **st)

    st = {}
    run(SMALL, False, 51, st, "par.")
    run(SMALL, True, 52, st, "cas.")
    np.savez_compressed(OUT / "coca_small.npz", **st)
    st = {}
    run(POOL96, False, 53, st, "par.")
    np.savez_compressed(OUT / "coca_pool96.npz", **st)

    with torch.device("meta"):
        big = coca_vit_l_14()
    sd = big.state_dict()
    np.savez_compressed(OUT / "coca_l14_meta.npz", keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]))
    print("coca fixtures written")


if __name__ == "__main__":
    main()
