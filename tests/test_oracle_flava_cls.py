"""Pin the classification-head / position-interpolation part of the numpy oracle to the reference
(fixture tests/golden/flava_cls_interp.npz from make_golden_flava_cls.py).  CPU-only."""
import numpy as np

from oracle import clip_oracle as oc
from tests._util import fixture_sd


def _hidden(sd_model, z, mode):
    lin = lambda name, x: x @ sd_model[name + ".weight"].T + sd_model[name + ".bias"]
    if mode == "image":
        return oc.flava_image_encoder(sd_model, "image_encoder.", z["image"], 2)["last_hidden_state"]
    if mode == "text":
        return oc.flava_text_encoder(sd_model, "text_encoder.", z["text"], 2)["last_hidden_state"]
    img = oc.flava_image_encoder(sd_model, "image_encoder.", z["image"], 2)
    txt = oc.flava_text_encoder(sd_model, "text_encoder.", z["text"], 2)
    fused = np.concatenate([lin("image_to_mm_projection", img["hidden_states"][-1]), lin("text_to_mm_projection", txt["hidden_states"][-1])], axis=1)
    return oc.flava_mm_encoder(sd_model, "mm_encoder.", fused, 2)["last_hidden_state"]


def test_classification_logits_and_loss(golden):
    """FLAVAForClassification.forward (models/flava/model.py:393-422) for the three embedding options and a non-zero cls_index."""
    z = golden("flava_cls_interp.npz")
    sd = fixture_sd(z)
    sd_model = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    for mode in ("image", "text", "mm"):
        h = _hidden(sd_model, z, mode)
        scores, loss = oc.flava_classification(sd, h, z["labels"], n_linear=2, stride=2)
        np.testing.assert_allclose(scores, z[mode + ".logits"], atol=3e-5, err_msg=mode)
        np.testing.assert_allclose(loss, z[mode + ".loss"], atol=2e-5, err_msg=mode)
    scores3, _ = oc.flava_classification(sd, _hidden(sd_model, z, "mm"), z["labels"], n_linear=2, stride=2, cls_index=3)
    np.testing.assert_allclose(scores3, z["mm.cls3.logits"], atol=3e-5)


def test_interpolated_position_embeddings(golden):
    """ImageEmbeddings.interpolate_pos_encoding (models/flava/image_encoder.py:102-137): torch's bicubic resampling restated."""
    z = golden("flava_cls_interp.npz")
    for side in (160, 96):
        n = (side // 16) ** 2
        got = oc.flava_interpolate_pos_encoding(z["interp.full_pos"].astype(np.float64), n, side, side, 16)
        assert got.shape == z[f"interp.full_{side}"].shape
        np.testing.assert_allclose(got, z[f"interp.full_{side}"], atol=2e-6)
    same = oc.flava_interpolate_pos_encoding(z["interp.full_pos"], 196, 224, 224, 16)
    assert same is z["interp.full_pos"] or np.array_equal(same, z["interp.full_pos"])
    # through the embeddings of the small model: 48x48 images on a model trained at 32x32 (2x2 grid -> 3x3)
    sd = fixture_sd(z)
    pre = "model.image_encoder."
    x = z["interp.image48"]
    w = sd[pre + "embeddings.patch_embeddings.projection.weight"]
    emb = oc.patch_embed(x, w) + sd[pre + "embeddings.patch_embeddings.projection.bias"]
    cls = np.broadcast_to(sd[pre + "embeddings.cls_token"].reshape(1, 1, -1), (2, 1, 128))
    pos = oc.flava_interpolate_pos_encoding(sd[pre + "embeddings.position_embeddings"], 9, 48, 48, 16)
    np.testing.assert_allclose(np.concatenate([cls, emb], axis=1) + pos, z["interp.emb48"], atol=2e-5)


def test_oracle_reproduces_the_reference_full_size_classification_kat():
    """tests/models/flava/test_flava.py:58-77 of the reference: seed 1234, inputs drawn first, flava_model_for_classification(2,
    pretrained=False).eval(): losses 0.7180 (mm) / 0.7020 (image) / 0.6663 (text) at atol 1e-4.  The oracle runs on the weights of the
    seeded host module (whose initialisation equals the reference's), so this pins oracle AND initialisation to the reference's own numbers."""
    import torch

    from multimodal_amd.models.flava.model import flava_model_for_classification
    from tests.conftest import set_rng_seed

    set_rng_seed(1234)
    text = torch.randint(0, 30500, (2, 77), dtype=torch.long).numpy()
    image = torch.rand((2, 3, 224, 224)).numpy()
    labels = torch.randint(0, 2, (2,), dtype=torch.long).numpy()
    model = flava_model_for_classification(2, pretrained=False)
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    sd_model = {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    lin = lambda name, x: x @ sd_model[name + ".weight"].T + sd_model[name + ".bias"]
    img = oc.flava_image_encoder(sd_model, "image_encoder.", image, 12)
    txt = oc.flava_text_encoder(sd_model, "text_encoder.", text, 12)
    fused = np.concatenate([lin("image_to_mm_projection", img["hidden_states"][-1]), lin("text_to_mm_projection", txt["hidden_states"][-1])], axis=1)
    mm = oc.flava_mm_encoder(sd_model, "mm_encoder.", fused, 12)
    for name, hidden, want in (("mm", mm["last_hidden_state"], 0.7180), ("image", img["last_hidden_state"], 0.7020), ("text", txt["last_hidden_state"], 0.6663)):
        _, loss = oc.flava_classification(sd, hidden, labels, n_linear=2, stride=3)
        assert abs(float(loss) - want) <= 1e-4, (name, float(loss), want)
