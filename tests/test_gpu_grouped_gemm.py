"""Grouped persistent GEMM (mmamd_gemm_bf16_grouped) and the layer-locked two-tower schedule built on it.

The grouped launch walks the concatenated tile lists of two problems; a tile's arithmetic does not depend on which launch computes
it, so the bar is BIT equality with one mmamd_gemm_bf16 call per problem (which the other GPU tests pin to the fp32 reference / the
oracle), for every epilogue kind the towers use, ragged M / N, in-place fp32 residuals, and the fall-back cases."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(M, N, K, seed, dev):
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(dev)
    return a, w, b


# (M, N, K) pairs: the cfg-2 tower shapes scaled down in M, ragged rows / columns, unequal K, the second problem larger than the first
SHAPES = [
    ((197 * 64, 2304, 768), (77 * 64, 1536, 512)),   # qkv of ViT-B/16 + text tower at B = 64 (49.25 / 19.25 row panels)
    ((197 * 48, 768, 3072), (77 * 80, 512, 2048)),   # MLP-down: long K, few column tiles
    ((77 * 64, 1544, 512), (197 * 64, 776, 768)),    # N not a multiple of 256 (nor of 64), smaller problem first
    ((16384, 1024, 128), (8192, 2048, 256)),         # shortest K the persistent kernel takes
]


@pytest.mark.parametrize("shapes", SHAPES)
@pytest.mark.parametrize("kind", ["bf16", "bf16_quickgelu", "bf16_gelu", "f32_residual", "bf16_residual", "f32"])
@torch.no_grad()
def test_grouped_equals_separate(shapes, kind):
    from multimodal_amd import ops

    dev = torch.device("cuda")
    act = {"bf16_quickgelu": ops.ACT_QUICKGELU, "bf16_gelu": ops.ACT_GELU_ERF}.get(kind, ops.ACT_NONE)
    odt = torch.float32 if kind.startswith("f32") else torch.bfloat16
    probs, want = [], []
    for i, (M, N, K) in enumerate(shapes):
        a, w, b = _mk(M, N, K, 100 + i, dev)
        res = None
        if kind.endswith("residual"):
            res = torch.randn(M, N, device=dev).to(odt)
        want.append(ops.gemm_bf16(a, w, b, act=act, residual=None if res is None else res.clone(), out_dtype=odt))
        out = res.clone() if res is not None else None  # residual aliases the output, as in the residual stream update
        probs.append((a, w, b, out, out))
    got = ops.gemm_bf16_grouped(probs, act=act, out_dtype=odt)
    for g, r in zip(got, want):
        assert g.dtype == odt and torch.equal(g, r)


@torch.no_grad()
def test_grouped_fallbacks_and_single_problem():
    from multimodal_amd import ops

    dev = torch.device("cuda")
    # K = 192 is not a multiple of 128, the second case has too few tiles for a persistent launch: both run as separate launches
    for shapes in [((8192, 512, 192), (4096, 512, 128)), ((512, 512, 256), (256, 256, 128))]:
        probs, want = [], []
        for i, (M, N, K) in enumerate(shapes):
            a, w, b = _mk(M, N, K, 7 + i, dev)
            want.append(ops.gemm_bf16(a, w, b))
            probs.append((a, w, b, None, None))
        for g, r in zip(ops.gemm_bf16_grouped(probs), want):
            assert torch.equal(g, r)
    # an empty problem next to a real one (a rank whose text batch is empty): nothing to launch for it, the other one is computed
    a, w, b = _mk(8192, 1024, 256, 5, dev)
    e = torch.empty((0, 256), dtype=torch.bfloat16, device=dev)
    got = ops.gemm_bf16_grouped([(a, w, b, None, None), (e, w, b, None, None)])
    assert got[1].shape == (0, 1024) and torch.equal(got[0], ops.gemm_bf16(a, w, b))
    a, w, b = _mk(4096, 768, 256, 3, dev)
    (g,) = ops.gemm_bf16_grouped([(a, w, b, None, None)])
    assert torch.equal(g, ops.gemm_bf16(a, w, b))
    with pytest.raises(ops.MmamdError):
        ops.gemm_bf16_grouped([])
    with pytest.raises(ops.MmamdError):
        ops.gemm_bf16_grouped([(a, w, b, None, None)] * 3)
    with pytest.raises(ops.MmamdError):
        ops.gemm_bf16_grouped([(a, w[:, :128].contiguous(), b, None, None)])


@pytest.mark.parametrize("factory,B", [("clip_vit_b16", 64), ("clip_vit_b32", 96), ("clip_vit_l14", 32)])
@torch.no_grad()
def test_grouped_two_tower_schedule_is_bit_identical_to_the_two_stream_one(factory, B, monkeypatch):
    """CLIP.forward: the layer-locked grouped schedule (default) and the two-stream schedule run the same arithmetic per tile.  L/14 has
    24 vision and 12 text layers: the upper 12 vision layers run alone."""
    from multimodal_amd.models import clip as clip_models
    from multimodal_amd.utils.synthetic import clip_batch

    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = getattr(clip_models, factory)().to(dev).eval()
    images, ids = clip_batch(B)
    images, ids = images.to(dev), ids.to(dev)
    from multimodal_amd.schedule import set_schedule

    prev = set_schedule(two_tower="streams")
    try:
        ref = model(images, ids)
        set_schedule(two_tower="grouped")
        got = model(images, ids)
    finally:
        set_schedule(two_tower=prev.two_tower)
    torch.cuda.synchronize()
    assert torch.equal(got.embeddings_a, ref.embeddings_a) and torch.equal(got.embeddings_b, ref.embeddings_b)
    assert torch.isfinite(got.embeddings_a).all() and got.embeddings_a.abs().sum() > 0
    # the packed [B, 2E] block contract of the loss's gather holds on both schedules
    E = got.embeddings_a.shape[1]
    assert got.embeddings_a.stride() == (2 * E, 1) and got.embeddings_b.data_ptr() == got.embeddings_a.data_ptr() + 4 * E


@torch.no_grad()
def test_grouped_schedule_keeps_hooks_and_errors():
    """A forward hook on an encoder must still fire (the grouped schedule bypasses the encoders' forward, so it steps aside), and a wrong
    text length still raises the reference's ValueError."""
    from multimodal_amd.models.clip import clip_vit_b32
    from multimodal_amd.utils.synthetic import clip_batch

    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = clip_vit_b32().to(dev).eval()
    images, ids = clip_batch(8)
    images, ids = images.to(dev), ids.to(dev)
    seen = []
    h = model.encoder_b.register_forward_hook(lambda m, i, o: seen.append(tuple(o.shape)))
    out_hooked = model(images, ids)
    h.remove()
    assert seen == [(8, 512)]
    assert torch.equal(model(images, ids).embeddings_b, out_hooked.embeddings_b)
    with pytest.raises(ValueError):
        model(images, ids[:, :50])
