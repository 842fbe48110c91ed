"""Pin the numpy oracle (oracle/clip_oracle.py) to the reference:
  (1) the reference's own hard-coded known-answer vectors for this path, and
  (2) outputs of the reference itself, generated in the build container (tests/golden/make_golden.py).
CPU-only; runs everywhere."""
import math

import numpy as np
import pytest
import torch

from oracle import clip_oracle as oc
from tests._util import assert_checksums, fixture_sd, sd_to_numpy
from tests.conftest import set_rng_seed


def test_quickgelu_kat():
    # reference tests/modules/layers/test_activation.py:12-16: silu(1) = 0.8458
    assert abs(float(oc.quick_gelu(np.array([1.0], dtype=np.float32))[0]) - 0.8458) < 1e-4


def test_vit_tiny_kat(golden):
    z = golden("kat_vit_tiny.npz")
    y = oc.clip_vit_forward(fixture_sd(z), z["x"], heads=2)
    # hard-coded in reference tests/models/clip/test_image_encoder.py:58-64
    np.testing.assert_allclose(y, np.array([[1.1296, -0.6523, 0.3949, -0.7351]] * 2), atol=1e-4, rtol=0)
    np.testing.assert_allclose(y, z["y"], atol=2e-6, rtol=0)


def test_text_hidden_state_kat(golden):
    z = golden("kat_text_hidden.npz")
    sd = fixture_sd(z)
    hs = oc.clip_text_forward(sd, z["text"], heads=2, return_hidden_state=True)
    # hard-coded in reference tests/models/clip/test_text_encoder.py:129-143
    expected = np.array([[[0.6348, -0.0414, -1.6042, 1.0108], [0.6205, -0.0303, -1.6066, 1.0164], [0.5916, -0.0017, -1.6133, 1.0234]],
                         [[0.5911, -0.0152, -1.6079, 1.0320], [0.1468, -1.6758, 0.7402, 0.7888], [0.6721, -0.2897, -1.4934, 1.1109]]])
    np.testing.assert_allclose(hs, expected, atol=1e-4, rtol=0)
    np.testing.assert_allclose(hs, z["hidden"], atol=5e-6, rtol=0)
    np.testing.assert_allclose(oc.clip_text_forward(sd, z["text"], heads=2), z["y"], atol=5e-6, rtol=0)


def test_text_full_kat_and_init_parity(golden):
    """Seeded construction of OUR module reproduces the reference's initial weights (checksums), and the oracle on
    those weights reproduces the reference's KAT (tests/models/clip/test_text_encoder.py:107-120)."""
    from multimodal_amd.models.clip import CLIPTextEncoder

    z = golden("kat_text_full.npz")
    set_rng_seed(1234)
    text = torch.randint(1, 10, (2, 77), dtype=torch.long)
    assert np.array_equal(text.numpy(), z["text"])
    enc = CLIPTextEncoder(embedding_dim=4, use_clip_init=True, context_length=77, width=512, heads=2)
    assert_checksums(enc, z)
    y = oc.clip_text_forward(sd_to_numpy(enc), text.numpy(), heads=2)
    np.testing.assert_allclose(y, np.array([[-1.3103, -0.6713, -0.9614, 0.7010], [1.1780, 0.1888, 0.8019, 0.7287]]), atol=1e-4, rtol=0)
    np.testing.assert_allclose(y, z["y"], atol=2e-5, rtol=0)


def test_loss_kats(golden):
    z = golden("loss_local.npz")
    o = oc.contrastive_loss_with_temperature(z["a"], z["b"], float(z["logit_scale"]))
    assert abs(float(o["loss"]) - 9.8753) < 1e-3  # reference test_contrastive_loss_with_temperature.py:75-82
    np.testing.assert_allclose(o["loss"], z["loss"], atol=1e-5)
    np.testing.assert_allclose(o["logits_a"], z["logits_a"], atol=1e-5)
    np.testing.assert_allclose(o["logits_b"], z["logits_b"], atol=1e-5)
    np.testing.assert_allclose(o["loss_a"], z["loss_a"], atol=1e-5)
    np.testing.assert_allclose(o["loss_b"], z["loss_b"], atol=1e-5)
    s = oc.contrastive_loss_with_temperature(z["a"], z["b"], float(z["logit_scale"]), label_smoothing=0.1)
    assert abs(float(s["loss"]) - 10.2524) < 1e-3  # reference :112-123
    np.testing.assert_allclose(s["loss"], z["loss_smooth"], atol=1e-5)
    m = oc.contrastive_loss_with_temperature(z["a"], z["b"], float(z["logit_scale"]), mask=z["mask"])
    np.testing.assert_allclose(m["loss"], z["loss_masked"], atol=1e-5)
    np.testing.assert_allclose(m["logits_a"], z["logits_a_masked"], atol=1e-5)


def test_clamp_semantics():
    assert oc.clamp_logit_scale(3.0, None, 2.0) == 2.0  # reference :84-110
    assert oc.clamp_logit_scale(1.0, 2.0, None) == 2.0
    assert oc.clamp_logit_scale(2.5, 0.0, 4.6052) == 2.5


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("bp", ["GLOBAL", "LOCAL", "NONE"])
def test_distributed_loss_matches_reference_gloo_run(golden, world, bp):
    z = golden("loss_dist.npz")
    a_all, b_all = z["a_all"], z["b_all"]
    B = a_all.shape[0] // world
    losses = []
    for r in range(world):
        o = oc.contrastive_loss_with_temperature(a_all[r * B:(r + 1) * B], b_all[r * B:(r + 1) * B], np.log(1 / 0.07),
                                                 a_all, b_all, rank=r)
        pre = f"w{world}.{bp}.r{r}."
        np.testing.assert_allclose(o["loss"], z[pre + "loss"], atol=2e-6)
        np.testing.assert_allclose(o["logits_a"], z[pre + "logits_a"], atol=2e-5)
        np.testing.assert_allclose(o["logits_b"], z[pre + "logits_b"], atol=2e-5)
        losses.append(float(o["loss"]))
    # mean over ranks == single-process loss on the concatenated batch (SURVEY.md §5 [probe])
    single = oc.contrastive_loss_with_temperature(a_all, b_all, np.log(1 / 0.07))
    assert abs(np.mean(losses) - float(single["loss"])) < 1e-5


def test_midsize_two_tower(golden):
    z = golden("midsize.npz")
    sd = fixture_sd(z)
    a, b = oc.clip_forward(sd, z["images"], z["ids"], vision_heads=2, text_heads=2)
    np.testing.assert_allclose(a, z["emb_a"], atol=2e-6)
    np.testing.assert_allclose(b, z["emb_b"], atol=2e-6)
    hid = oc.clip_text_forward(sd, z["ids"], heads=2, prefix="encoder_b.", return_hidden_state=True)
    np.testing.assert_allclose(hid, z["text_hidden"], atol=2e-5)
    o = oc.contrastive_loss_with_temperature(a, b, np.log(1 / 0.07))
    np.testing.assert_allclose(o["loss"], z["loss"], atol=1e-5)
    np.testing.assert_allclose(o["logits_a"], z["logits_a"], atol=5e-5)


@pytest.mark.parametrize("name,factory,B,vh", [("clip_b32_b8", "clip_vit_b32", 8, 12), ("clip_b16_b4", "clip_vit_b16", 4, 12)])
def test_full_size_clip_against_reference_run(golden, name, factory, B, vh):
    """cfg 1 (ViT-B/32, B=8) and the cfg-2 model (ViT-B/16) at B=4: seed-0 default init + the shared synthetic batch."""
    import multimodal_amd.models.clip as mc
    from multimodal_amd.utils.synthetic import clip_batch

    z = golden(name + ".npz")
    set_rng_seed(0)
    model = getattr(mc, factory)()
    assert_checksums(model, z)
    images, ids = clip_batch(B)
    assert abs(float(images.double().sum()) - float(z["images_sum"])) < 1e-6 and int(ids.sum()) == int(z["ids_sum"])
    sd = sd_to_numpy(model)
    a, b = oc.clip_forward(sd, images.numpy(), ids.numpy(), vision_heads=vh, text_heads=8)
    np.testing.assert_allclose(a, z["emb_a"], atol=1e-5)
    np.testing.assert_allclose(b, z["emb_b"], atol=1e-5)
    o = oc.contrastive_loss_with_temperature(a, b, np.log(1 / 0.07))
    np.testing.assert_allclose(o["logits_a"], z["logits_a"], atol=2e-4)
    np.testing.assert_allclose(o["loss"], z["loss"], atol=1e-4)
    assert np.array_equal(o["logits_a"].argmax(1), z["logits_a"].argmax(1))


@pytest.mark.parametrize("name,factory,B", [("clip_b32_b8", "clip_vit_b32", 8), ("clip_b16_b4", "clip_vit_b16", 4)])
def test_torch_cpu_restatement_matches_reference_fixtures(golden, name, factory, B):
    """oracle/torch_cpu_clip.py (what bench.py's cpu_baseline leg times on the GPU box, where /root/reference is absent) is the reference's
    own module composition on the same torch.nn modules: its outputs equal the reference's at fp32 round-off."""
    import multimodal_amd.models.clip as mc
    from multimodal_amd.utils.synthetic import clip_batch
    from oracle.torch_cpu_clip import TorchCPUCLIP

    z = golden(name + ".npz")
    set_rng_seed(0)
    model = getattr(mc, factory)()
    assert_checksums(model, z)
    torch.set_num_threads(8)
    m = TorchCPUCLIP(model.state_dict(), vision_heads=12, text_heads=8)
    images, ids = clip_batch(B)
    a, b, la, lb, loss = m.forward_loss(images, ids)
    np.testing.assert_allclose(a.numpy(), z["emb_a"], atol=2e-5)
    np.testing.assert_allclose(b.numpy(), z["emb_b"], atol=2e-5)
    np.testing.assert_allclose(la.numpy(), z["logits_a"], atol=5e-4)
    np.testing.assert_allclose(lb.numpy(), z["logits_b"], atol=5e-4)
    assert abs(float(loss) - float(z["loss"])) < 2e-5


def test_headline_fixture_records_the_reference_bf16_cpu_rates(golden):
    """clip_b16_b256.npz carries the reference's own bf16-CPU argmax agreement (97.3 % image->text at B = 256), the bar
    tests/test_gpu_headline_parity.py holds the HIP path to; its loss is the one committed in profiles/r02_reference_cpu.json."""
    import json
    from pathlib import Path

    z = golden("clip_b16_b256.npz")
    assert z["emb_a"].shape == (256, 512) and z["logits_a"].shape == (256, 256) and z["bf16_logits_b"].shape == (256, 256)
    ra = float((z["bf16_logits_a"].argmax(1) == z["logits_a"].argmax(1)).mean())
    assert ra == float(z["bf16_agree_a"]) and 0.9 < ra <= 1.0
    np.testing.assert_allclose(np.linalg.norm(z["emb_a"], axis=1), 1.0, atol=1e-5)
    t = math.exp(math.log(1 / 0.07))
    np.testing.assert_allclose(z["emb_a"].astype(np.float64) @ z["emb_b"].astype(np.float64).T * t, z["logits_a"], atol=2e-4)
    o = oc.contrastive_loss_with_temperature(z["emb_a"], z["emb_b"], math.log(1 / 0.07))
    assert abs(float(o["loss"]) - float(z["loss"])) < 1e-5  # the oracle's loss arithmetic at the headline size
    rec = json.loads((Path(__file__).resolve().parents[1] / "profiles" / "r02_reference_cpu.json").read_text())["clip_b16_b256"]
    assert abs(rec["loss"] - float(z["loss"])) < 1e-6 and rec["batch"] == 256
