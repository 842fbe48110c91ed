"""Model-level parity on an MI355X: the drop-in nn.Modules (multimodal_amd.models.clip / modules.losses) against
(a) outputs of the reference itself (committed fixtures, tests/golden/make_golden.py) and (b) the numpy oracle.

Tolerances (SURVEY.md §8c protocol; the HIP path computes GEMMs/attention from bf16 operands with fp32 accumulation,
the reference/oracle is fp32 end to end):  embeddings |d| <= 4e-3, logits |d| <= 0.06 at T = 1/0.07, loss |d| <= 5e-3,
argmax identical on every row whose fp32 top-1/top-2 margin exceeds 2 x the logit tolerance.
"""
import math

import numpy as np
import pytest
import torch

from oracle import clip_oracle as oc
from tests._util import assert_checksums, fixture_sd
from tests.conftest import set_rng_seed

pytestmark = pytest.mark.gpu

EMB_TOL, LOGIT_TOL, LOSS_TOL = 4e-3, 0.06, 5e-3


@pytest.fixture(scope="module", autouse=True)
def _built():
    from multimodal_amd import build

    build.build()


def host(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def check_argmax(logits_hip, logits_ref):
    top2 = np.sort(logits_ref, axis=1)[:, -2:]
    safe = (top2[:, 1] - top2[:, 0]) > 2 * LOGIT_TOL
    assert np.array_equal(logits_hip.argmax(1)[safe], logits_ref.argmax(1)[safe])
    return float((logits_hip.argmax(1) == logits_ref.argmax(1)).mean())


def test_midsize_two_tower_vs_reference_fixture(golden):
    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import (
        ContrastiveLossWithTemperature, contrastive_loss_with_temperature)

    z = golden("midsize.npz")
    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=2, patch_size=16, image_size=64, width=128)
    txt = CLIPTextEncoder(embedding_dim=64, context_length=77, vocab_size=1000, width=128, dim_feedforward=256, heads=2, layers=2)
    clip = CLIP(vit, txt)
    clip.load_state_dict({k: torch.from_numpy(v) for k, v in fixture_sd(z).items()}, strict=True)
    clip = clip.cuda().eval()
    images, ids = torch.from_numpy(z["images"]).cuda(), torch.from_numpy(z["ids"]).cuda()
    with torch.no_grad():
        out = clip(images, ids)
        hid = clip.encoder_b(ids, return_hidden_state=True)
        loss_mod = ContrastiveLossWithTemperature().cuda()
        lo = contrastive_loss_with_temperature(out.embeddings_a, out.embeddings_b, loss_mod.logit_scale)
        loss = loss_mod(out.embeddings_a, out.embeddings_b)
    assert isinstance(out, tuple) and out._fields == ("embeddings_a", "embeddings_b")
    np.testing.assert_allclose(host(out.embeddings_a), z["emb_a"], atol=EMB_TOL)
    np.testing.assert_allclose(host(out.embeddings_b), z["emb_b"], atol=EMB_TOL)
    assert hid.shape == (6, 77, 128)
    np.testing.assert_allclose(host(hid), z["text_hidden"], atol=3e-2)
    np.testing.assert_allclose(host(lo.logits_a), z["logits_a"], atol=LOGIT_TOL)
    np.testing.assert_allclose(host(lo.logits_b), z["logits_b"], atol=LOGIT_TOL)
    assert abs(float(lo.loss) - float(z["loss"])) < LOSS_TOL and abs(float(loss) - float(z["loss"])) < LOSS_TOL
    check_argmax(host(lo.logits_a), z["logits_a"].astype(np.float64))
    # and the same against the oracle evaluated here on the same inputs
    a, b = oc.clip_forward(fixture_sd(z), z["images"], z["ids"], 2, 2)
    np.testing.assert_allclose(host(out.embeddings_a), a, atol=EMB_TOL)
    np.testing.assert_allclose(host(out.embeddings_b), b, atol=EMB_TOL)


@pytest.mark.parametrize("name,factory,B", [("clip_b32_b8", "clip_vit_b32", 8), ("clip_b16_b4", "clip_vit_b16", 4)])
def test_full_size_clip_vs_reference_fixture(golden, name, factory, B):
    import multimodal_amd.models.clip as mc
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import contrastive_loss_with_temperature
    from multimodal_amd.utils.synthetic import clip_batch

    z = golden(name + ".npz")
    set_rng_seed(0)
    model = getattr(mc, factory)()
    assert_checksums(model, z)
    model = model.cuda().eval()
    images, ids = clip_batch(B)
    with torch.no_grad():
        out = model(images.cuda(), ids.cuda())
        raw_a = model.encoder_a(images.cuda())
        raw_b = model.encoder_b(ids.cuda())
        scale = torch.nn.Parameter(torch.tensor(math.log(1 / 0.07), device="cuda"))
        lo = contrastive_loss_with_temperature(out.embeddings_a, out.embeddings_b, scale)
    np.testing.assert_allclose(host(out.embeddings_a), z["emb_a"], atol=EMB_TOL)
    np.testing.assert_allclose(host(out.embeddings_b), z["emb_b"], atol=EMB_TOL)
    # un-normalised tower outputs: relative check (their scale is model dependent)
    for got, ref in ((host(raw_a), z["raw_a"]), (host(raw_b), z["raw_b"])):
        assert np.abs(got - ref).max() <= 2e-2 * np.abs(ref).max()
    np.testing.assert_allclose(host(lo.logits_a), z["logits_a"], atol=LOGIT_TOL)
    np.testing.assert_allclose(host(lo.logits_b), z["logits_b"], atol=LOGIT_TOL)
    assert abs(float(lo.loss) - float(z["loss"])) < LOSS_TOL
    check_argmax(host(lo.logits_a), z["logits_a"].astype(np.float64))
    check_argmax(host(lo.logits_b), z["logits_b"].astype(np.float64))


def test_headline_config_properties_b256():
    """cfg 2 at full size (ViT-B/16, B=256): size-independent properties instead of a 35-s CPU oracle run.
      * batch-composition invariance: rows 0..7 of the B=256 run == a B=8 run of the same rows, BIT-exact
        (every kernel's per-row arithmetic order is independent of the row's tile position);
      * unit-norm embeddings; loss == oracle loss recomputed from the HIP embeddings; logits block == a.b^T*T.
      * oracle parity on a 4-row slice (those rows ARE part of the big batch)."""
    import multimodal_amd.models.clip as mc
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import contrastive_loss_with_temperature
    from multimodal_amd.utils.synthetic import clip_batch

    set_rng_seed(0)
    model = mc.clip_vit_b16()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    model = model.cuda().eval()
    images, ids = clip_batch(256)
    with torch.no_grad():
        big = model(images.cuda(), ids.cuda())
        small = model(images[:8].cuda(), ids[:8].cuda())
        scale = torch.nn.Parameter(torch.tensor(math.log(1 / 0.07), device="cuda"))
        lo = contrastive_loss_with_temperature(big.embeddings_a, big.embeddings_b, scale)
    assert torch.equal(big.embeddings_a[:8], small.embeddings_a) and torch.equal(big.embeddings_b[:8], small.embeddings_b)
    a, b = host(big.embeddings_a), host(big.embeddings_b)
    np.testing.assert_allclose(np.linalg.norm(a, axis=1), 1.0, atol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(b, axis=1), 1.0, atol=1e-5)
    ref = oc.contrastive_loss_with_temperature(a, b, math.log(1 / 0.07), dtype=np.float64)
    np.testing.assert_allclose(host(lo.logits_a), ref["logits_a"], atol=5e-5)
    assert abs(float(lo.loss) - float(ref["loss"])) < 1e-4
    oa, ob = oc.clip_forward(sd, images[:4].numpy(), ids[:4].numpy(), 12, 8)
    np.testing.assert_allclose(a[:4], oa, atol=EMB_TOL)
    np.testing.assert_allclose(b[:4], ob, atol=EMB_TOL)


def test_clip_vit_l14_vs_oracle():
    """cfg-3 model (ViT-L/14: width 1024, 16 heads, S=257, patch 14 -> K padded 588->640, E=768) at B=2 vs the oracle."""
    import multimodal_amd.models.clip as mc
    from multimodal_amd.utils.synthetic import clip_batch

    set_rng_seed(0)
    model = mc.clip_vit_l14()
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    model = model.cuda().eval()
    images, ids = clip_batch(2)
    with torch.no_grad():
        out = model(images.cuda(), ids.cuda())
    a, b = oc.clip_forward(sd, images.numpy(), ids.numpy(), 16, 12)
    assert out.embeddings_a.shape == (2, 768)
    np.testing.assert_allclose(host(out.embeddings_a), a, atol=EMB_TOL)
    np.testing.assert_allclose(host(out.embeddings_b), b, atol=EMB_TOL)


def test_bf16_parameters_and_generic_towers():
    """model.to(bfloat16) keeps working (outputs follow the parameter dtype) and CLIP stays tower-agnostic
    (reference tests/models/clip/test_clip.py:26-56 uses arbitrary towers)."""
    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder

    set_rng_seed(3)
    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=1, patch_size=16, image_size=32, width=128)
    txt = CLIPTextEncoder(embedding_dim=64, context_length=16, vocab_size=100, width=128, dim_feedforward=256, heads=2, layers=1)
    clip32 = CLIP(vit, txt).cuda().eval()
    images = torch.randn(3, 3, 32, 32).cuda()
    ids = torch.randint(1, 100, (3, 16)).cuda()
    with torch.no_grad():
        o32 = clip32(images, ids)
        import copy

        clip16 = copy.deepcopy(clip32).to(torch.bfloat16)
        o16 = clip16(images.to(torch.bfloat16), ids)
    assert o16.embeddings_a.dtype == torch.bfloat16 and o32.embeddings_a.dtype == torch.float32
    np.testing.assert_allclose(host(o16.embeddings_a), host(o32.embeddings_a), atol=3e-2)

    class Tower(torch.nn.Module):
        def __init__(self, t):
            super().__init__()
            self.t = t

        def forward(self, x):
            return self.t

    ta, tb = torch.randn(5, 7).cuda(), torch.randn(5, 7).cuda()
    out = CLIP(Tower(ta), Tower(tb))(None, None)
    np.testing.assert_allclose(host(out.embeddings_a), oc.l2_normalize(host(ta)), atol=1e-6)


def test_loss_module_semantics_on_device(golden):
    """In-place clamp of the parameter, clamp equivalences, label smoothing, mask (reference loss tests :75-127)."""
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import (
        ContrastiveLossWithTemperature, contrastive_loss_with_temperature)

    z = golden("loss_local.npz")
    a, b = torch.from_numpy(z["a"]).cuda(), torch.from_numpy(z["b"]).cuda()
    with torch.no_grad():
        mod = ContrastiveLossWithTemperature().cuda()
        assert abs(float(mod(a, b)) - 9.8753) < 1e-3
        assert abs(float(mod(a, b, cross_entropy_kwargs={"label_smoothing": 0.1})) - 10.2524) < 1e-3
        at_max = ContrastiveLossWithTemperature(logit_scale=2, logit_scale_max=2).cuda()
        above = ContrastiveLossWithTemperature(logit_scale=3, logit_scale_max=2).cuda()
        assert abs(float(at_max(a, b)) - float(above(a, b))) < 1e-3
        assert abs(float(above.logit_scale) - 2.0) < 1e-6  # the parameter itself was mutated
        at_min = ContrastiveLossWithTemperature(logit_scale=2, logit_scale_min=2).cuda()
        below = ContrastiveLossWithTemperature(logit_scale=1, logit_scale_min=2).cuda()
        assert abs(float(at_min(a, b)) - float(below(a, b))) < 1e-3
        mask = torch.from_numpy(z["mask"]).cuda()
        lo = contrastive_loss_with_temperature(a, b, mod.logit_scale, mask=mask)
    assert lo.logits_a.shape == tuple(z["logits_a_masked"].shape)
    np.testing.assert_allclose(host(lo.logits_a), z["logits_a_masked"], atol=1e-4)
    assert abs(float(lo.loss) - float(z["loss_masked"])) < 1e-4
    import dataclasses

    assert [f.name for f in dataclasses.fields(lo)] == ["loss", "logits_a", "logits_b", "loss_a", "loss_b"]


def test_global_loss_over_rccl_single_rank(golden, tmp_path):
    """The N>1 code path (packed all_gather_into_tensor over the 'nccl' = RCCL backend) on the one GPU of the box."""
    import torch.distributed as dist

    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import contrastive_loss_with_temperature
    from multimodal_amd.utils.distributed import BackpropType

    z = golden("loss_dist.npz")
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/sync", world_size=1, rank=0)
    try:
        a, b = torch.from_numpy(z["a_all"]).cuda(), torch.from_numpy(z["b_all"]).cuda()
        scale = torch.nn.Parameter(torch.tensor(math.log(1 / 0.07), device="cuda"))
        with torch.no_grad():
            for bp in BackpropType:
                lo = contrastive_loss_with_temperature(a, b, scale, backprop_type=bp)
                assert abs(float(lo.loss) - float(z[f"w1.{bp.name}.r0.loss"])) < 1e-4
                np.testing.assert_allclose(host(lo.logits_a), z[f"w1.{bp.name}.r0.logits_a"], atol=1e-4)
    finally:
        dist.destroy_process_group()


def test_forward_is_graph_capturable_and_replays_bit_identically():
    """The C-ABI neither allocates nor synchronises: the two-tower forward + loss can be captured into one HIP graph
    (torch.cuda.CUDAGraph) and replayed on new inputs copied into the captured buffers."""
    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    set_rng_seed(7)
    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=2, patch_size=16, image_size=64, width=128)
    txt = CLIPTextEncoder(embedding_dim=64, context_length=77, vocab_size=1000, width=128, dim_feedforward=256, heads=2, layers=2)
    clip = CLIP(vit, txt).cuda().eval()
    loss_fn = ContrastiveLossWithTemperature().cuda()
    im1, id1 = clip_batch(6, image_size=64, vocab_size=1000, rank=1)
    im2, id2 = clip_batch(6, image_size=64, vocab_size=1000, rank=2)
    s_im, s_id = im1.cuda().clone(), id1.cuda().clone()
    with torch.no_grad():
        for _ in range(2):  # warm-up: parameter packing, lazy handles
            o = clip(s_im, s_id)
            loss_fn(o.embeddings_a, o.embeddings_b)
        eager = []
        for im, idt in ((im1, id1), (im2, id2)):
            o = clip(im.cuda(), idt.cuda())
            eager.append((o.embeddings_a.clone(), o.embeddings_b.clone(), loss_fn(o.embeddings_a, o.embeddings_b).clone()))
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                o = clip(s_im, s_id)
                l = loss_fn(o.embeddings_a, o.embeddings_b)
        torch.cuda.current_stream().wait_stream(side)
        for (im, idt), (ea, eb, el) in zip(((im1, id1), (im2, id2)), eager):
            s_im.copy_(im.cuda()); s_id.copy_(idt.cuda())
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(o.embeddings_a, ea) and torch.equal(o.embeddings_b, eb) and torch.equal(l, el)


def test_ninety_six_wide_heads_on_the_clip_stack_vs_oracle():
    """CLIP encoders whose heads are not 64 wide (VERDICT r03 missing #6; the reference takes any width / heads): 96-wide heads run the
    general attention kernel on the packed projection's column blocks -- ViT (non-causal) and text tower (causal) vs the numpy oracle."""
    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder

    set_rng_seed(5)
    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=2, patch_size=16, image_size=32, width=192)
    txt = CLIPTextEncoder(embedding_dim=64, context_length=16, vocab_size=100, width=192, dim_feedforward=256, heads=2, layers=2)
    model = CLIP(vit, txt)
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    model = model.cuda().eval()
    images = torch.randn(3, 3, 32, 32)
    ids = torch.randint(1, 99, (3, 16))
    ids[:, 5] = 99  # EOT = arg-max id
    with torch.no_grad():
        out = model(images.cuda(), ids.cuda())
    a, b = oc.clip_forward(sd, images.numpy(), ids.numpy(), 2, 2)
    np.testing.assert_allclose(host(out.embeddings_a), a, atol=EMB_TOL)
    np.testing.assert_allclose(host(out.embeddings_b), b, atol=EMB_TOL)
