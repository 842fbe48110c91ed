"""LDS-DMA ring attention kernel (csrc/attention_ring.hip, the default of mmamd_attention_fwd for S <= 208) against float64 math, against
the r02 register-staged kernel it replaces (bit for bit on non-causal problems: same arithmetic in the same order), in grouped
two-problem launches, with many more items than workgroups (ring wrap-around) and with the log-sum-exp output.  Needs an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    from multimodal_amd import build

    build.build()


def _ref(qkv: torch.Tensor, B, S, H, causal):
    x = qkv.double().cpu().numpy().reshape(B, S, 3, H, 64)
    q, k, v = (x[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(0, 1, 3, 2) / 8.0
    if causal:
        s = np.where(np.triu(np.ones((S, S), dtype=bool), 1), -np.inf, s)
    mx = s.max(-1, keepdims=True)
    p = np.exp(s - mx)
    den = p.sum(-1, keepdims=True)
    lse2 = (mx + np.log(den))[..., 0] * 1.4426950408889634  # log2-domain log-sum-exp of the scaled scores
    return ((p / den) @ v).transpose(0, 2, 1, 3).reshape(B * S, H * 64), lse2


def _old_kernel(fn):
    from multimodal_amd import _lib

    _lib.lib().mmamd_debug_set_attn_variant(1000)
    try:
        return fn()
    finally:
        _lib.lib().mmamd_debug_set_attn_variant(0)


SHAPES = [(3, 197, 2, False), (2, 77, 8, True), (5, 50, 3, False), (1, 1, 1, False), (2, 7, 1, True), (2, 33, 2, True), (2, 64, 1, False),
          (1, 208, 2, False), (1, 208, 1, True), (1, 224, 2, False), (2, 200, 1, False), (3, 193, 1, False), (2, 129, 2, True), (4, 96, 2, False), (1, 31, 1, True)]


@pytest.mark.parametrize("B,S,H,causal", SHAPES)
def test_ring_vs_float64_and_old_kernel(B, S, H, causal):
    from multimodal_amd import ops

    g = torch.Generator().manual_seed(S * 7 + H)
    qkv = (torch.randn(B * S, 3 * H * 64, generator=g) * 1.5).to(torch.bfloat16).cuda()
    out = ops.attention_fwd(qkv, B, S, H, causal)
    ref, _ = _ref(qkv, B, S, H, causal)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=3e-2, rtol=2e-2)
    old = _old_kernel(lambda: ops.attention_fwd(qkv, B, S, H, causal))
    if not causal:  # the pipelined key loop of the old kernel is non-causal only; the causal one sums the row in another order
        assert torch.equal(out, old)
    else:
        assert (out.float() - old.float()).abs().max().item() <= 4e-2


def test_ring_many_items_per_workgroup_and_garbage_beyond_items():
    """More items than 256 workgroups x ring slots (the ring wraps several times) and a qkv tensor whose memory around the items is
    poisoned with NaN-free huge values: rows >= S of a slot alias other items' data and must not reach the output."""
    from multimodal_amd import ops

    g = torch.Generator().manual_seed(3)
    for B, S, H, causal in ((96, 197, 12, False), (300, 77, 8, True), (70, 50, 12, False)):
        qkv = (torch.randn(B * S, 3 * H * 64, generator=g) * 2.0).to(torch.bfloat16).cuda()
        out = ops.attention_fwd(qkv, B, S, H, causal)
        old = _old_kernel(lambda: ops.attention_fwd(qkv, B, S, H, causal))
        if causal:
            assert (out.float() - old.float()).abs().max().item() <= 4e-2
        else:
            assert torch.equal(out, old)
        # a 4-batch slice against float64
        ref, _ = _ref(qkv[: 4 * S], 4, S, H, causal)
        np.testing.assert_allclose(out[: 4 * S].float().cpu().numpy(), ref, atol=3e-2, rtol=2e-2)


def test_ring_grouped_equals_separate_launches():
    from multimodal_amd import ops

    g = torch.Generator().manual_seed(9)
    Ba, Sa, Ha, Bb, Sb, Hb = 40, 197, 12, 40, 77, 8
    qa = torch.randn(Ba * Sa, 3 * Ha * 64, generator=g).to(torch.bfloat16).cuda()
    qb = torch.randn(Bb * Sb, 3 * Hb * 64, generator=g).to(torch.bfloat16).cuda()
    oa = ops.attention_fwd(qa, Ba, Sa, Ha, False)
    ob = ops.attention_fwd(qb, Bb, Sb, Hb, True)
    ga, gb = ops.attention_fwd_grouped([(qa, Ba, Sa, Ha, False, None), (qb, Bb, Sb, Hb, True, None)])
    assert torch.equal(oa, ga) and torch.equal(ob, gb)
    # the other order, and a first problem with fewer items than workgroups
    gb2, ga2 = ops.attention_fwd_grouped([(qb, Bb, Sb, Hb, True, None), (qa, Ba, Sa, Ha, False, None)])
    assert torch.equal(oa, ga2) and torch.equal(ob, gb2)
    qs = qa[: 2 * Sa].contiguous()
    gs, gb3 = ops.attention_fwd_grouped([(qs, 2, Sa, Ha, False, None), (qb, Bb, Sb, Hb, True, None)])
    assert torch.equal(gs, oa[: 2 * Sa]) and torch.equal(gb3, ob)
    # a problem the ring kernel does not take (S = 257) falls back to separate launches
    ql = torch.randn(2 * 257, 3 * 2 * 64, generator=g).to(torch.bfloat16).cuda()
    gl, gb4 = ops.attention_fwd_grouped([(ql, 2, 257, 2, False, None), (qb, Bb, Sb, Hb, True, None)])
    assert torch.equal(gl, ops.attention_fwd(ql, 2, 257, 2, False)) and torch.equal(gb4, ob)


def test_ring_spiked_key_forces_the_deferred_rescale():
    from multimodal_amd import ops

    rng = np.random.default_rng(5)
    S = 197
    qkv = rng.standard_normal((S, 192)) * 0.3
    qkv[10, 0:64] = 4.0      # query 10
    qkv[170, 64:128] = 4.0   # key 170 (tile 5): score 128 >> the others
    t = torch.from_numpy(qkv).to(torch.bfloat16).cuda()
    out = ops.attention_fwd(t, 1, S, 1, False).float().cpu().numpy()
    np.testing.assert_allclose(out[10], t[170, 128:192].float().cpu().numpy(), atol=2e-2)


@pytest.mark.parametrize("B,S,H,causal", [(3, 197, 2, False), (4, 77, 2, True), (2, 50, 1, False)])
def test_ring_log_sum_exp(B, S, H, causal):
    from multimodal_amd import ops

    g = torch.Generator().manual_seed(S)
    qkv = torch.randn(B * S, 3 * H * 64, generator=g).to(torch.bfloat16).cuda()
    out, lse = ops.attention_fwd_train(qkv, B, S, H, causal)
    ref, lse_ref = _ref(qkv, B, S, H, causal)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=3e-2, rtol=2e-2)
    np.testing.assert_allclose(lse.cpu().numpy().reshape(B, H, S), lse_ref, atol=2e-3, rtol=1e-4)


def test_add_layernorm_grouped_vs_float64():
    from multimodal_amd import ops

    g = torch.Generator().manual_seed(4)
    xa, xb = torch.randn(777, 768, generator=g).cuda(), (torch.randn(130, 512, generator=g) * 3 + 1).cuda()
    da, db = torch.randn(777, 768, generator=g).bfloat16().cuda(), torch.randn(130, 512, generator=g).bfloat16().cuda()
    ga, ba = torch.randn(768, generator=g).cuda(), torch.randn(768, generator=g).cuda()
    gb, bb = torch.randn(512, generator=g).cuda(), torch.randn(512, generator=g).cuda()
    ya, yb = torch.empty_like(da), torch.empty_like(db)

    def ref(x, d, gm, bt, eps):
        xn = x.double() + (d.double() if d is not None else 0)
        mu, var = xn.mean(-1, keepdim=True), xn.var(-1, unbiased=False, keepdim=True)
        return xn, (xn - mu) / torch.sqrt(var + eps) * gm.double() + bt.double()

    # plain LayerNorm of both problems == two mmamd_layernorm launches, bit for bit; x untouched
    xa0, xb0 = xa.clone(), xb.clone()
    ops.add_layernorm_grouped([(xa, None, ga, ba, 1e-5, ya), (xb, None, gb, bb, 1e-6, yb)])
    assert torch.equal(xa, xa0) and torch.equal(xb, xb0)
    assert torch.equal(ya, ops.layernorm(xa, ga, ba, 1e-5)) and torch.equal(yb, ops.layernorm(xb, gb, bb, 1e-6))
    # with deltas: x updated in place, y = LN(x + delta)
    rxa, rya = ref(xa, da, ga, ba, 1e-5)
    rxb, ryb = ref(xb, db, gb, bb, 1e-6)
    ops.add_layernorm_grouped([(xa, da, ga, ba, 1e-5, ya), (xb, db, gb, bb, 1e-6, yb)])
    assert (xa.double() - rxa).abs().max().item() <= 1e-6 and (xb.double() - rxb).abs().max().item() <= 2e-6
    np.testing.assert_allclose(ya.double().cpu().numpy(), rya.cpu().numpy(), atol=1e-2, rtol=8e-3)  # bf16 output
    np.testing.assert_allclose(yb.double().cpu().numpy(), ryb.cpu().numpy(), atol=1e-2, rtol=8e-3)
    # add only (no LayerNorm output), one problem
    x1 = xa.clone()
    ops.add_layernorm_grouped([(x1, da, None, None, 0.0, None)])
    assert torch.equal(x1, xa + da.float())
