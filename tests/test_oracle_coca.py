"""The numpy oracle's CoCa restatement (oracle/clip_oracle.py: coca_*) pinned to the reference: its constant-weight KAT
(tests/models/coca/test_coca_model.py: 0.3536 / 8.0 / 0.6931 / 3.9120) and outputs of the reference itself on seeded random
small models (tests/golden/make_golden_coca.py).  Weights of the random models are re-created from the seed through the drop-in
modules and verified against the reference's per-tensor checksums."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as oc
from tests._util import assert_checksums, fixture_sd, sd_to_numpy
from tests.golden.make_golden import seed
from tests.golden.make_golden_coca import POOL96, randomize, SMALL


def rebuild(kw, cascaded, seed_v, z, prefix):
    from multimodal_amd.models.coca.coca_model import coca_vit

    seed(seed_v)
    model = coca_vit(**kw, cascaded_pooler=cascaded).eval()
    randomize(model, torch.Generator().manual_seed(seed_v + 1))
    assert_checksums(model, {"keys": z[prefix + "keys"], "sums": z[prefix + "sums"], "asums": z[prefix + "asums"]})
    return model


def test_coca_kat_constant_weights(golden):
    z = golden("coca_kat.npz")
    sd = fixture_sd(z)
    out = oc.coca_model_forward(sd, z["images"], z["texts"], 2, 2, 2, 2, cascaded=False)
    assert np.abs(out["image_pooled_output"] - 0.3536).max() <= 1e-4 and np.abs(out["text_pooled_output"] - 0.3536).max() <= 1e-4
    assert np.abs(out["multimodal_embeddings"] - 8.0).max() <= 1e-4
    for k in ("image_pooled_output", "text_pooled_output", "multimodal_embeddings"):
        assert np.abs(out[k] - z[k]).max() <= 1e-5, k
    losses = oc.coca_pretraining_losses(out, z["texts"], float(z["logit_scale"]))
    assert abs(float(losses["contrastive"]) - 0.6931) <= 1e-4 and abs(float(losses["captioning"]) - 3.9120) <= 1e-4


@pytest.mark.parametrize("fixture,kw,cascaded,seed_v,prefix,heads", [
    ("coca_small.npz", SMALL, False, 51, "par.", (2, 2, 2, 2)), ("coca_small.npz", SMALL, True, 52, "cas.", (2, 2, 2, 2)),
    ("coca_pool96.npz", POOL96, False, 53, "par.", (2, 3, 3, 2))])
def test_coca_oracle_vs_reference_outputs(golden, fixture, kw, cascaded, seed_v, prefix, heads):
    z = golden(fixture)
    model = rebuild(kw, cascaded, seed_v, z, prefix)
    out = oc.coca_model_forward(sd_to_numpy(model), z[prefix + "images"], z[prefix + "texts"], *heads, cascaded=cascaded)
    for k, tol in (("image_pooled_output", 2e-6), ("text_pooled_output", 2e-6), ("multimodal_embeddings", 5e-5)):
        assert out[k].shape == z[prefix + k].shape, k
        assert np.abs(out[k] - z[prefix + k]).max() <= tol, (k, np.abs(out[k] - z[prefix + k]).max())
    if not cascaded:
        losses = oc.coca_pretraining_losses(out, z[prefix + "texts"], np.log(1 / 0.07))
        assert abs(float(losses["contrastive"]) - float(z[prefix + "loss_contrastive"])) <= 2e-5
        assert abs(float(losses["captioning"]) - float(z[prefix + "loss_captioning"])) <= 2e-5


def test_coca_text_mask_matches_reference_build_mask():
    """build_mask semantics restated independently with torch ops (F.pad shifts the padding mask by one column)."""
    import torch.nn.functional as F

    ids = torch.tensor([[5, 3, 0, 0], [1, 2, 3, 4], [0, 7, 0, 9]])
    pm = (ids != 0).unsqueeze(1)
    ref = (F.pad(pm, (1, 0, pm.shape[2], 0), value=1.0) * torch.tril(torch.ones(5, 5)).bool()).unsqueeze(1)
    assert np.array_equal(oc.coca_text_mask(ids.numpy(), 0), ref.bool().numpy())
