"""Kernel-level parity: every libmmamd.so entry point (called through the C-ABI via multimodal_amd.ops) against
the numpy oracle / a float64 restatement of the same op on the same seeded inputs.  Needs an MI355X (-m gpu)."""
import math

import numpy as np
import pytest
import torch

from oracle import clip_oracle as oc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    from multimodal_amd import build

    build.build()


def dev():
    return torch.device("cuda", 0)


def bf16_round(x: np.ndarray) -> np.ndarray:
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def to_dev(x: np.ndarray, dtype=torch.float32) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev()).to(dtype).contiguous()


def host(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy().astype(np.float64)


# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,d", [(1, 64), (5, 512), (197 * 3, 768), (33, 1024), (7, 2048)])
@pytest.mark.parametrize("in_dtype,out_dtype", [(torch.float32, torch.bfloat16), (torch.float32, torch.float32),
                                                (torch.bfloat16, torch.bfloat16)])
def test_layernorm(rows, d, in_dtype, out_dtype):
    from multimodal_amd import ops

    rng = np.random.default_rng(rows * 1000 + d)
    x = rng.standard_normal((rows, d)).astype(np.float32) * 3 + 0.5
    g = rng.standard_normal(d).astype(np.float32)
    b = rng.standard_normal(d).astype(np.float32)
    xt = to_dev(x, in_dtype)
    y = ops.layernorm(xt, to_dev(g), to_dev(b), 1e-5, out_dtype=out_dtype)
    ref = oc.layer_norm(host(xt), g.astype(np.float64), b.astype(np.float64), 1e-5)
    tol = 2e-5 if out_dtype == torch.float32 else 4e-2
    np.testing.assert_allclose(host(y), ref, atol=tol, rtol=1e-2 if out_dtype != torch.float32 else 1e-5)


# ----------------------------------------------------------------------------------------------
GEMM_SHAPES = [
    (300, 256, 64), (591, 384, 128), (256, 256, 768), (1000, 768, 768), (77 * 5, 1536, 512), (130, 3072, 768),
    (197 * 4, 768, 3072), (33, 128, 64), (1, 64, 64),
]


@pytest.mark.parametrize("variant", [0, 1, 2, 5, 6, 7, 18])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_f32_out_exactness(variant, M, N, K):
    """fp32 output: bf16 inputs are exact in fp32, so the only error is fp32 accumulation order."""
    from multimodal_amd import ops

    rng = np.random.default_rng(M * 7 + N * 3 + K)
    a = bf16_round(rng.standard_normal((M, K)))
    w = bf16_round(rng.standard_normal((N, K)) * 0.5 + 0.1 * np.arange(N)[:, None] / N)  # asymmetric
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    ops.set_gemm_variant(variant)
    try:
        rt = to_dev(res)
        c = ops.gemm_bf16(to_dev(a, torch.bfloat16), to_dev(w, torch.bfloat16), to_dev(bias), residual=rt, out=rt,
                          out_dtype=torch.float32)
    finally:
        ops.set_gemm_variant(0)
    ref = a.astype(np.float64) @ w.astype(np.float64).T + bias + res
    np.testing.assert_allclose(host(c), ref, atol=2e-3 * math.sqrt(K / 64), rtol=1e-5)


@pytest.mark.parametrize("variant", [0, 2, 5, 7, 18, 99])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_bf16_out_epilogues(variant, act):
    from multimodal_amd import ops

    M, N, K = 450, 512, 256
    rng = np.random.default_rng(act + 10 * variant)
    a = bf16_round(rng.standard_normal((M, K)) * 0.5)
    w = bf16_round(rng.standard_normal((N, K)) * 0.2)
    bias = rng.standard_normal(N).astype(np.float32)
    res = bf16_round(rng.standard_normal((M, N)))
    ops.set_gemm_variant(variant)
    try:
        c = ops.gemm_bf16(to_dev(a, torch.bfloat16), to_dev(w, torch.bfloat16), to_dev(bias), act=act,
                          residual=to_dev(res, torch.bfloat16))
        c2 = ops.gemm_bf16(to_dev(a, torch.bfloat16), to_dev(w, torch.bfloat16), None, act=act)
    finally:
        ops.set_gemm_variant(0)
    z = a.astype(np.float64) @ w.astype(np.float64).T

    def f(v):
        if act == 1:
            return v / (1 + np.exp(-1.702 * v))
        if act == 2:
            from scipy.special import erf

            return 0.5 * v * (1 + erf(v / math.sqrt(2)))
        return v

    np.testing.assert_allclose(host(c), f(z + bias) + res, atol=3e-2, rtol=1e-2)
    np.testing.assert_allclose(host(c2), f(z), atol=3e-2, rtol=1e-2)


def test_gemm_argument_errors():
    from multimodal_amd import ops

    a = torch.zeros((4, 48), dtype=torch.bfloat16, device=dev())
    w = torch.zeros((8, 48), dtype=torch.bfloat16, device=dev())
    with pytest.raises(ops.MmamdError):
        ops.gemm_bf16(a, w)  # K not a multiple of 64
    with pytest.raises(ops.MmamdError):
        ops.gemm_bf16(a.cpu(), w.cpu())


# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,S,H,causal", [(3, 50, 2, False), (2, 77, 8, True), (2, 197, 12, False), (1, 257, 3, False),
                                          (2, 64, 1, True), (2, 33, 2, True), (1, 1, 1, False), (1, 275, 2, False),
                                          (1, 288, 1, True),
                                          # S > 288: the streaming (chunked K/V) kernel — 384-pixel ViT (577), 512-token BERT, ragged tails
                                          (2, 577, 3, False), (1, 512, 2, True), (1, 289, 1, False), (2, 401, 2, True)])
def test_attention(B, S, H, causal):
    from multimodal_amd import ops

    rng = np.random.default_rng(S * 13 + H)
    D = H * 64
    qkv = bf16_round(rng.standard_normal((B * S, 3 * D)) * 1.5)
    out = ops.attention_fwd(to_dev(qkv, torch.bfloat16), B, S, H, causal)
    x = qkv.astype(np.float64).reshape(B, S, 3, H, 64)
    q, k, v = (x[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(0, 1, 3, 2) / 8.0
    if causal:
        s = np.where(np.triu(np.ones((S, S), dtype=bool), 1), -np.inf, s)
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    ref = (p @ v).transpose(0, 2, 1, 3).reshape(B * S, D)
    np.testing.assert_allclose(host(out), ref, atol=3e-2, rtol=2e-2)


def test_attention_spiked_row_forces_running_max_update():
    """One key (in a LATE tile) dominates one query: the running-max rescale branch must be exact (guide rule 26)."""
    from multimodal_amd import ops

    rng = np.random.default_rng(5)
    B, S, H = 1, 197, 1
    qkv = rng.standard_normal((S, 192)) * 0.3
    qkv[10, 0:64] = 4.0           # query 10
    qkv[170, 64:128] = 4.0        # key 170 (tile 5): score 16*64/8 = 128 >> others
    qkv = bf16_round(qkv)
    out = host(ops.attention_fwd(to_dev(qkv, torch.bfloat16), B, S, H, False))
    np.testing.assert_allclose(out[10], qkv[170, 128:192], atol=2e-2)


# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("P,HW,dtype", [(16, 64, torch.float32), (32, 64, torch.float32), (14, 56, torch.float32),
                                        (16, 224, torch.bfloat16)])
def test_patchify_and_patch_embed(P, HW, dtype):
    from multimodal_amd import ops

    rng = np.random.default_rng(P)
    B, w = 2, 128
    img = rng.standard_normal((B, 3, HW, HW)).astype(np.float32)
    it = to_dev(img, dtype)
    K = 3 * P * P
    kpad = (K + 63) // 64 * 64
    patches = ops.patchify(it, P, kpad)
    g = HW // P
    ref = host(it).reshape(B, 3, g, P, g, P).transpose(0, 2, 4, 1, 3, 5).reshape(B * g * g, K)
    got = host(patches)
    np.testing.assert_array_equal(got[:, :K], bf16_round(ref).astype(np.float64))
    assert (got[:, K:] == 0).all()
    conv_w = bf16_round(rng.standard_normal((w, 3, P, P)) * 0.05)
    wp = np.zeros((w, kpad), dtype=np.float32)
    wp[:, :K] = conv_w.reshape(w, K)
    pe = ops.gemm_bf16(patches, to_dev(wp, torch.bfloat16), out_dtype=torch.float32)
    ref_pe = oc.patch_embed(bf16_round(host(it)).astype(np.float64), conv_w.astype(np.float64)).reshape(B * g * g, w)
    np.testing.assert_allclose(host(pe), ref_pe, atol=2e-3, rtol=1e-4)


def test_vit_assemble_ln():
    from multimodal_amd import ops

    rng = np.random.default_rng(1)
    B, G2, d = 3, 49, 768
    pe = rng.standard_normal((B * G2, d)).astype(np.float32)
    cls, pos = rng.standard_normal(d).astype(np.float32), rng.standard_normal((G2 + 1, d)).astype(np.float32)
    g, b = rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)
    x = ops.vit_assemble_ln(to_dev(pe), to_dev(cls), to_dev(pos), to_dev(g), to_dev(b), 1e-5, B, G2)
    full = np.concatenate([np.broadcast_to(cls, (B, 1, d)), pe.reshape(B, G2, d)], axis=1).astype(np.float64) + pos
    ref = oc.layer_norm(full, g.astype(np.float64), b.astype(np.float64), 1e-5).reshape(B * (G2 + 1), d)
    np.testing.assert_allclose(host(x), ref, atol=3e-5, rtol=1e-5)


@pytest.mark.parametrize("table_dtype", [torch.float32, torch.bfloat16])
def test_embed_tokens(table_dtype):
    from multimodal_amd import ops

    rng = np.random.default_rng(2)
    B, S, d, V = 4, 77, 512, 1000
    table = to_dev(rng.standard_normal((V, d)).astype(np.float32), table_dtype)
    pos = rng.standard_normal((S, d)).astype(np.float32)
    ids = rng.integers(0, V, (B, S))
    x = ops.embed_tokens(torch.from_numpy(ids).to(dev()), table, to_dev(pos))
    ref = host(table)[ids].reshape(B * S, d) + np.tile(pos, (B, 1))
    np.testing.assert_allclose(host(x), ref, atol=1e-6)


@pytest.mark.parametrize("linear_weight,with_ids,normalize", [(False, False, False), (True, True, False), (True, True, True),
                                                              (False, False, True)])
def test_pool_ln_proj(linear_weight, with_ids, normalize):
    from multimodal_amd import ops

    rng = np.random.default_rng(3)
    B, S, d, E = 5, 77, 512, 256
    x = rng.standard_normal((B * S, d)).astype(np.float32)
    g, b = rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)
    proj = (rng.standard_normal((E, d) if linear_weight else (d, E)) * 0.05).astype(np.float32)
    ids = rng.integers(1, 400, (B, S))
    ids[0, 5] = ids[0, 40] = 999  # tie: first maximum wins (torch.argmax semantics)
    idx = ids.argmax(1) if with_ids else np.zeros(B, dtype=int)
    out = ops.pool_ln_proj(to_dev(x), B, S, torch.from_numpy(ids).to(dev()) if with_ids else None, to_dev(g), to_dev(b),
                           1e-5, to_dev(proj), linear_weight, normalize)
    rows = x.reshape(B, S, d)[np.arange(B), idx].astype(np.float64)
    h = oc.layer_norm(rows, g.astype(np.float64), b.astype(np.float64), 1e-5)
    ref = h @ (proj.T if linear_weight else proj).astype(np.float64)
    if normalize:
        ref = oc.l2_normalize(ref)
    np.testing.assert_allclose(host(out), ref, atol=2e-5, rtol=1e-5)


def test_l2_normalize_and_clamp_and_convert():
    from multimodal_amd import ops

    rng = np.random.default_rng(4)
    x = rng.standard_normal((9, 512)).astype(np.float32)
    x[3] = 0.0  # eps branch: 0 / max(0, eps) = 0
    y = ops.l2_normalize(to_dev(x))
    np.testing.assert_allclose(host(y), oc.l2_normalize(x.astype(np.float64)), atol=1e-6)
    for v, lo, hi, want in [(3.0, None, 2.0, 2.0), (1.0, 2.0, None, 2.0), (2.5, 0.0, 4.6052, 2.5), (9.0, 0.0, 4.6052, 4.6052)]:
        p = torch.tensor([v], device=dev())
        ops.clamp_scalar_(p, lo, hi)
        assert abs(float(p) - want) < 1e-6
    c = ops.convert(to_dev(x), torch.bfloat16)
    np.testing.assert_array_equal(host(c), bf16_round(x).astype(np.float64))
    np.testing.assert_array_equal(host(ops.convert(c, torch.float32)), bf16_round(x).astype(np.float64))


# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,W,E,rank", [(3, 1, 5, 0), (16, 1, 24, 0), (4, 4, 24, 2), (256, 1, 512, 0), (70, 3, 100, 1), (33, 2, 768, 1)])
@pytest.mark.parametrize("smoothing", [0.0, 0.1])
def test_contrastive_fwd(B, W, E, rank, smoothing):
    from multimodal_amd import ops

    rng = np.random.default_rng(B + E)
    a_all = oc.l2_normalize(rng.standard_normal((W * B, E))).astype(np.float32)
    b_all = oc.l2_normalize(rng.standard_normal((W * B, E))).astype(np.float32)
    buf = to_dev(np.concatenate([a_all, b_all], axis=1))
    a, b = a_all[rank * B:(rank + 1) * B], b_all[rank * B:(rank + 1) * B]
    ls = torch.tensor([math.log(1 / 0.07)], device=dev())
    mask = rng.random(B) > 0.3 if B > 3 else None
    mt = None if mask is None else torch.from_numpy(mask).to(dev()).view(torch.uint8)
    out3, la, lb = ops.contrastive_fwd(to_dev(a), to_dev(b), buf[:, :E], buf[:, E:], 2 * E, ls, B * rank, mt, smoothing)
    ref = oc.contrastive_loss_with_temperature(a, b, math.log(1 / 0.07), a_all, b_all, rank, None, smoothing, dtype=np.float64)
    np.testing.assert_allclose(host(la), ref["logits_a"], atol=2e-5)
    np.testing.assert_allclose(host(lb), ref["logits_b"], atol=2e-5)
    if mask is not None:
        ref = oc.contrastive_loss_with_temperature(a, b, math.log(1 / 0.07), a_all, b_all, rank, mask, smoothing, dtype=np.float64)
    got = host(out3)
    np.testing.assert_allclose(got, [ref["loss"], ref["loss_a"], ref["loss_b"]], atol=2e-5, rtol=1e-5)


def test_contrastive_fwd_reference_kats(golden):
    """The reference's own loss known answers (tests/modules/losses/test_contrastive_loss_with_temperature.py:75-123)."""
    from multimodal_amd import ops

    z = golden("loss_local.npz")
    a, b = to_dev(z["a"]), to_dev(z["b"])
    ls = torch.tensor([float(z["logit_scale"])], device=dev())
    out3, la, lb = ops.contrastive_fwd(a, b, a, b, 5, ls, 0)
    assert abs(float(out3[0]) - 9.8753) < 1e-3
    np.testing.assert_allclose(host(la), z["logits_a"], atol=1e-4)
    out3s, _, _ = ops.contrastive_fwd(a, b, a, b, 5, ls, 0, None, 0.1)
    assert abs(float(out3s[0]) - 10.2524) < 1e-3


@torch.no_grad()
def test_attention_probs_pipelined_equals_serial_key_loops():
    """attention_probs_fwd: the software-pipelined key loops (default) and the serial ones (mmamd_debug_set_attn_variant(512)) run the same
    arithmetic per row — outputs and probabilities bit-identical, with and without a key-padding mask, fp32 and bf16 probabilities."""
    from multimodal_amd import _lib, ops

    g = torch.Generator().manual_seed(11)
    try:
        for B, S, H in ((3, 197, 2), (2, 77, 4), (2, 275, 2), (2, 20, 1)):
            qkv = torch.randn(B * S, 3 * H * 64, generator=g).to(torch.bfloat16).cuda()
            km = (torch.rand(B, S, generator=g) > 0.3).to(torch.uint8)
            km[:, 0] = 1
            for mask in (None, km.cuda()):
                for dt in (torch.float32, torch.bfloat16):
                    _lib.lib().mmamd_debug_set_attn_variant(512)
                    o0, p0 = ops.attention_probs_fwd(qkv, B, S, H, mask, want_probs=True, probs_dtype=dt)
                    _lib.lib().mmamd_debug_set_attn_variant(513)
                    o1, p1 = ops.attention_probs_fwd(qkv, B, S, H, mask, want_probs=True, probs_dtype=dt)
                    assert torch.equal(o0, o1) and torch.equal(p0, p1), (B, S, H, mask is not None, dt)
    finally:
        _lib.lib().mmamd_debug_set_attn_variant(513)
