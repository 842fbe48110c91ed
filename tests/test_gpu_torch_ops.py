"""The reference's scripting tests on an MI355X (row b of SURVEY.md section 8): tests/models/clip/test_text_encoder.py:162-174,
tests/modules/layers/test_multi_head_attention.py:50-57 — a scripted module must return what the eager module returns — plus
torch.compile of the CLIP model through the dispatcher ops of csrc/torch_ops.cpp (torch.ops.mmamd.*)."""
import numpy as np
import pytest
import torch

from tests.conftest import set_rng_seed

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    from multimodal_amd import _torch_ops, build

    build.build()
    _torch_ops.load()


def test_scripted_text_encoder_equals_eager_and_reference(golden):
    """reference test_text_encoder.py:162-174 scripts CLIPTextEncoder and compares with known answers.  Its KAT model (width 512 on 2
    heads = 256-wide heads) is outside the MI355X attention kernels (64-wide heads; the eager forward refuses it too), so the scripted
    module is checked on the kernel-legal text tower of tests/golden/midsize.npz: bit-identical to the eager forward, and within the
    usual tolerance of the reference's own output."""
    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder
    from tests._util import fixture_sd

    z = golden("midsize.npz")
    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=2, patch_size=16, image_size=64, width=128)
    txt = CLIPTextEncoder(embedding_dim=64, context_length=77, vocab_size=1000, width=128, dim_feedforward=256, heads=2, layers=2)
    clip = CLIP(vit, txt)
    clip.load_state_dict({k: torch.from_numpy(v) for k, v in fixture_sd(z).items()}, strict=True)
    enc = clip.encoder_b.cuda().eval()
    text = torch.from_numpy(z["ids"]).cuda()
    scripted = torch.jit.script(enc)
    with torch.no_grad():
        actual, eager = scripted(text), enc(text)
        hid_s, hid_e = scripted(text, True), enc(text, return_hidden_state=True)
    assert torch.equal(actual, eager) and torch.equal(hid_s, hid_e)  # same kernels, same arithmetic: scripted == eager, bit for bit
    assert np.abs(hid_s.float().cpu().numpy() - z["text_hidden"]).max() <= 3e-2
    nb = torch.nn.functional.normalize(actual.float().cpu(), dim=1).numpy()
    assert np.abs(nb - z["emb_b"]).max() <= 4e-3
    with pytest.raises((ValueError, torch.jit.Error), match="length of input should be 77"):
        scripted(text[:, :76])
    # the reference's KAT configuration itself: refused identically by both forwards (head width 256)
    kat = CLIPTextEncoder(embedding_dim=4, use_clip_init=True, context_length=77, width=512, heads=2).cuda().eval()
    with pytest.raises(Exception, match="64"), torch.no_grad():
        kat(text.clamp(max=9))
    with pytest.raises(Exception, match="64"):
        torch.jit.script(kat)(text.clamp(max=9))


def test_scripted_multi_head_self_attention_equals_eager():
    """reference test_multi_head_attention.py:50-57 (there: embed_dim 4; the MI355X kernels are built for 64-wide heads)."""
    from multimodal_amd.modules.layers.multi_head_attention import MultiHeadSelfAttention

    set_rng_seed(4)
    mha = MultiHeadSelfAttention(128, 2).cuda().eval().requires_grad_(False)
    q = torch.randn(3, 50, 128, device="cuda")
    scripted = torch.jit.script(mha)
    assert torch.equal(scripted(q), mha(q))
    assert torch.equal(scripted(q, None, True), mha(q, is_causal=True))
    with pytest.raises(Exception, match="attn_mask"):
        scripted(q, torch.ones(50, 50, dtype=torch.bool, device="cuda"))


def _small_clip():
    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder

    set_rng_seed(3)
    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=2, patch_size=16, image_size=64, width=128)
    txt = CLIPTextEncoder(embedding_dim=64, context_length=77, vocab_size=1000, width=128, dim_feedforward=256, heads=2, layers=2)
    return CLIP(vit, txt).cuda().eval()


def test_scripted_and_compiled_clip_equal_eager():
    from multimodal_amd.modules.layers.activation import SiLU
    from multimodal_amd.utils.synthetic import clip_batch

    clip = _small_clip()
    images, ids = clip_batch(5, image_size=64, vocab_size=1000)
    images, ids = images.cuda(), ids.cuda()
    with torch.no_grad():
        eager = clip(images, ids)
        scripted = torch.jit.script(clip)(images, ids)
        assert torch.equal(scripted.embeddings_a, eager.embeddings_a) and torch.equal(scripted.embeddings_b, eager.embeddings_b)
        # torch.compile: dynamo traces the module into ONE graph of torch.ops.mmamd.* calls (fullgraph: no graph break, i.e. no opaque
        # ctypes call was hit), FakeTensor propagation runs on the Meta kernels of the shim; aot_eager executes the real ops
        for backend in ("aot_eager", "inductor"):  # inductor (torch.compile's default) emits the custom ops as extern calls
            torch._dynamo.reset()
            got = torch.compile(clip, backend=backend, fullgraph=True)(images, ids)
            assert torch.equal(got.embeddings_a, eager.embeddings_a) and torch.equal(got.embeddings_b, eager.embeddings_b), backend
        x = torch.randn(7, 33, device="cuda")
        assert torch.equal(torch.jit.script(SiLU())(x), SiLU()(x))


def test_ops_pack_parameters_and_follow_updates():
    """`packed` keeps one kernel-ready copy per parameter and refreshes it when the parameter changes in place."""
    ns = torch.ops.mmamd
    w = torch.nn.Parameter(torch.randn(64, 128, device="cuda"))
    a = torch.randn(10, 128, device="cuda").to(torch.bfloat16)
    y0 = ns.gemm_bf16(a, w, None, None, 0, 0)
    assert ns.packed(w, 1).data_ptr() == ns.packed(w, 1).data_ptr()  # cached
    ref = a.float() @ w.detach().to(torch.bfloat16).float().t()
    assert (y0 - ref).abs().max() < 1e-3 * ref.abs().max() + 1e-3
    with torch.no_grad():
        w.mul_(2.0)  # in-place: version counter moves
    assert torch.equal(ns.gemm_bf16(a, w, None, None, 0, 0), 2 * y0)


@pytest.mark.parametrize("fixture,kw_name,cascaded,seed_v,prefix", [
    ("coca_small.npz", "SMALL", False, 51, "par."), ("coca_small.npz", "SMALL", True, 52, "cas."), ("coca_pool96.npz", "POOL96", False, 53, "par.")])
def test_scripted_coca_model_equals_eager(golden, fixture, kw_name, cascaded, seed_v, prefix):
    """reference tests/models/coca/test_coca_model.py:146-154: scripted_model(images, texts) == coca_model(images, texts) (atol 1e-4).  The
    reference's own KAT model (3- and 4-wide heads) is outside the MI355X attention kernels; the kernel-legal small models of
    tests/golden/make_golden_coca.py are used instead, whose eager outputs are pinned to the reference by tests/test_gpu_coca.py."""
    from tests.golden import make_golden_coca as mg
    from tests.test_gpu_coca import rebuild

    z = golden(fixture)
    model = rebuild(getattr(mg, kw_name), cascaded, seed_v, z, prefix).cuda()
    images, texts = torch.from_numpy(z[prefix + "images"]).cuda(), torch.from_numpy(z[prefix + "texts"]).cuda()
    scripted = torch.jit.script(model)
    with torch.no_grad():
        eager = model(images, texts)
        actual = scripted(images, texts)
        actual_pm = scripted(images, texts, texts != 0)  # explicit padding mask == the default (ids != pad)
    for k in ("image_pooled_output", "text_pooled_output", "multimodal_embeddings"):
        a, e = getattr(actual, k), getattr(eager, k)
        assert a.shape == e.shape and a.dtype == e.dtype, k
        assert float((a - e).abs().max()) <= 1e-4, (k, float((a - e).abs().max()))  # the reference's tolerance
        assert torch.equal(getattr(actual_pm, k), a), k
    # the pooled outputs run the very same kernels in both forms
    assert torch.equal(actual.image_pooled_output, eager.image_pooled_output)


def test_compiled_coca_model_equals_scripted():
    """torch.compile(coca, fullgraph=True) traces the same dispatcher-op forwards (FakeTensor shapes from the Meta kernels)."""
    from multimodal_amd.models.coca.coca_model import coca_vit
    from tests.golden.make_golden_coca import SMALL

    set_rng_seed(7)
    model = coca_vit(**SMALL, cascaded_pooler=False).cuda().eval()
    g = torch.Generator().manual_seed(3)
    images = torch.randn(2, 3, SMALL.get("image_size", 224), SMALL.get("image_size", 224), generator=g).cuda()
    texts = torch.randint(1, SMALL["vocab_size"], (2, SMALL["num_text_positions"]), generator=g)
    texts[0, 7:] = 0
    texts = texts.cuda()
    scripted = torch.jit.script(model)
    with torch.no_grad():
        want = scripted(images, texts)
    for backend in ("aot_eager", "inductor"):
        torch._dynamo.reset()
        compiled = torch.compile(model, backend=backend, fullgraph=True)
        with torch.no_grad():
            got = compiled(images, texts)
        for k in ("image_pooled_output", "text_pooled_output", "multimodal_embeddings"):
            assert torch.equal(getattr(got, k), getattr(want, k)), (backend, k)


@torch.no_grad()
def test_attn_probs_op_equals_the_eager_kernel_call():
    """torch.ops.mmamd.attn_probs (attn_fwd(..., write_probs)) == ops.attention_probs_fwd: FLAVA's attention with the probabilities."""
    from multimodal_amd import _torch_ops, ops

    ns = _torch_ops.load()
    g = torch.Generator().manual_seed(2)
    B, S, H = 3, 197, 2
    qkv = torch.randn(B * S, 3 * H * 64, generator=g).to(torch.bfloat16).cuda()
    km = (torch.rand(B, S, generator=g) > 0.2).to(torch.uint8)
    km[:, 0] = 1
    km = km.cuda()
    for mask in (None, km):
        for dt, code in ((torch.float32, 0), (torch.bfloat16, 1)):
            o_ref, p_ref = ops.attention_probs_fwd(qkv, B, S, H, mask, want_probs=True, probs_dtype=dt)
            o, p = ns.attn_probs(qkv, B, S, H, mask, True, code)
            assert torch.equal(o, o_ref) and torch.equal(p, p_ref) and p.dtype == dt
    o, p = ns.attn_probs(qkv, B, S, H, None, False, 0)
    assert torch.equal(o, ops.attention_probs_fwd(qkv, B, S, H, None, want_probs=False)[0]) and p.numel() == 0


def test_invalidate_packed_reaches_the_shim_cache():
    """`.data` writes + invalidate_packed() must also refresh the C++ packed-parameter cache behind the scripted forwards."""
    from multimodal_amd import _torch_ops
    from multimodal_amd._packing import invalidate_packed

    ns = _torch_ops.load()
    w = torch.nn.Parameter(torch.randn(8, 64, device="cuda"))
    a = ns.packed(w, 1).clone()
    w.data.mul_(2.0)  # no version bump
    assert torch.equal(ns.packed(w, 1), a)  # stale by construction ...
    invalidate_packed()
    assert torch.equal(ns.packed(w, 1), (w.detach()).to(torch.bfloat16))  # ... until invalidated
