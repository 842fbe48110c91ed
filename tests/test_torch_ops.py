"""The TORCH_LIBRARY(mmamd, ...) shim (csrc/torch_ops.cpp) and the scriptable forwards built on it — CPU part: the library builds /
loads, every op is registered with the expected schema, the Meta kernels infer shapes and dtypes (what FakeTensor tracing under
torch.compile uses), the modules the reference scripts in its tests (tests/models/clip/test_text_encoder.py:162-174,
tests/modules/layers/test_multi_head_attention.py:50-57) compile with torch.jit.script, keep their ValueError, and fail loudly — not
silently on some fallback — when handed CPU tensors."""
import pytest
import torch

OPS = ["abi_version", "packed", "convert", "layernorm", "gemm_bf16", "attn_fwd", "patch_embed", "embed_tokens", "pool_proj_normalize",
       "l2_normalize", "clamp_scalar_", "activation", "contrastive_fwd"]


@pytest.fixture(scope="module")
def ns():
    from multimodal_amd import _lib, _torch_ops

    n = _torch_ops.load()
    assert n.abi_version() == _lib.ABI_VERSION
    return n


def test_ops_are_registered_with_schemas(ns):
    for name in OPS:
        op = getattr(ns, name)
        assert callable(op)
    s = str(torch.ops.mmamd.gemm_bf16.default._schema)
    assert "Tensor a, Tensor w, Tensor? bias, Tensor? residual, int act, int out_dtype" in s
    assert "Tensor(a!) p" in str(torch.ops.mmamd.clamp_scalar_.default._schema)


def test_meta_kernels_infer_shapes_and_dtypes(ns):
    bf, f32 = torch.bfloat16, torch.float32
    m = lambda *s, dtype=f32: torch.empty(*s, dtype=dtype, device="meta")  # noqa: E731
    assert ns.gemm_bf16(m(10, 64, dtype=bf), m(24, 64), m(24), None, 1, 1).shape == (10, 24)
    o = ns.gemm_bf16(m(10, 64, dtype=bf), m(24, 64), None, m(10, 24), 0, 0)
    assert o.dtype == f32 and o.shape == (10, 24)
    assert ns.layernorm(m(6, 128), m(128), m(128), 1e-5, 1).dtype == bf
    assert ns.attn_fwd(m(2 * 77, 3 * 128, dtype=bf), 2, 77, 2, True).shape == (154, 128)
    x = ns.patch_embed(m(3, 3, 64, 64), m(128, 3, 16, 16), m(128), m(17, 128), m(128), m(128), 1e-5, 16)
    assert x.shape == (3 * 17, 128) and x.dtype == f32
    assert ns.embed_tokens(torch.empty(2, 77, dtype=torch.int64, device="meta"), m(1000, 128), m(77, 128)).shape == (154, 128)
    assert ns.pool_proj_normalize(m(154, 128), 2, 77, None, m(128), m(128), 1e-5, m(64, 128), True, False).shape == (2, 64)
    assert ns.pool_proj_normalize(m(154, 128), 2, 77, None, m(128), m(128), 1e-5, m(128, 32), False, True).shape == (2, 32)
    out3, la, lb = ns.contrastive_fwd(m(4, 16), m(4, 16), m(8, 16), m(8, 16), m(1), 4, None, 0.0, 0)
    assert out3.shape == (3,) and la.shape == (4, 8) and lb.shape == (4, 8)
    assert ns.packed(m(5, 7), 1).dtype == bf and ns.l2_normalize(m(4, 16), 1e-12).shape == (4, 16)


def test_reference_scripted_modules_compile_and_keep_their_errors(ns):
    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.modules.layers.activation import SiLU
    from multimodal_amd.modules.layers.multi_head_attention import MultiHeadSelfAttention
    from multimodal_amd.modules.layers.normalizations import Fp32LayerNorm

    txt = CLIPTextEncoder(embedding_dim=4, context_length=77, width=128, heads=2, layers=2, dim_feedforward=256, vocab_size=100)
    vit = CLIPViTEncoder(embedding_dim=4, heads=2, layers=1, patch_size=16, image_size=32, width=128)
    scripted = torch.jit.script(txt)
    assert isinstance(scripted, torch.jit.ScriptModule)
    code = scripted.code + scripted._forward_ops.code + scripted.encoder.code
    for op in ("embed_tokens", "layernorm", "gemm_bf16", "attn_fwd", "pool_proj_normalize"):
        assert f"ops.mmamd.{op}" in code, op  # the scripted graph calls the HIP kernels' dispatcher ops, nothing else computes
    with pytest.raises((ValueError, torch.jit.Error), match="length of input should be 77"):
        scripted(torch.zeros(2, 76, dtype=torch.long))  # reference test_text_encoder.py:151-160 through the scripted module
    with pytest.raises(Exception, match="mmamd|CPU|backend"):  # CPU tensors: no kernel for the CPU backend -> loud, no fallback
        scripted(torch.zeros(2, 77, dtype=torch.long))
    sv = torch.jit.script(vit)
    with pytest.raises((ValueError, torch.jit.Error), match="Expected 3 channels"):
        sv(torch.zeros(1, 1, 32, 32))
    assert "ops.mmamd.l2_normalize" in torch.jit.script(CLIP(vit, txt)).code
    assert "ops.mmamd.attn_fwd" in torch.jit.script(MultiHeadSelfAttention(128, 2))._forward_ops.code  # reference test_multi_head_attention.py:50-57
    assert "ops.mmamd.activation" in torch.jit.script(SiLU()).code
    assert "ops.mmamd.layernorm" in torch.jit.script(Fp32LayerNorm(16)).code
