"""Backward kernels on an MI355X against float64 restatements of what torch autograd computes."""
import numpy as np
import pytest
import torch

from tests.conftest import set_rng_seed

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    from multimodal_amd import build

    build.build()


def host(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("B,S,H,causal", [(2, 16, 1, False), (3, 77, 8, True), (2, 197, 12, False), (1, 33, 2, True), (2, 257, 2, False),
                                          (1, 288, 1, True), (2, 1, 1, False)])
def test_attention_backward(B, S, H, causal):
    from multimodal_amd import ops

    set_rng_seed(S + 13 * H)
    D = H * 64
    qkv = torch.randn(B * S, 3 * D).to(torch.bfloat16)
    dout = torch.randn(B * S, D).to(torch.bfloat16)
    out, lse = ops.attention_fwd_train(qkv.cuda(), B, S, H, causal)
    x = qkv.float().numpy().astype(np.float64).reshape(B, S, 3, H, 64)
    q, k, v = (x[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(0, 1, 3, 2) / 8.0
    if causal:
        s = np.where(np.tril(np.ones((S, S), dtype=bool)), s, -np.inf)
    m = s.max(-1, keepdims=True)
    p = np.exp(s - m)
    l = p.sum(-1, keepdims=True)
    p /= l
    o = p @ v
    ref_lse = (m[..., 0] + np.log(l[..., 0])) / np.log(2.0)
    assert np.abs(host(lse) - ref_lse).max() <= 1e-4
    assert np.abs(host(out) - o.transpose(0, 2, 1, 3).reshape(B * S, D)).max() <= 2e-2 * max(1.0, np.abs(o).max())
    # the forward kernel rounds O to bf16 and the backward reads that: use the same O for Dq in the reference
    o_used = host(out).reshape(B, S, H, 64).transpose(0, 2, 1, 3)
    do = dout.float().numpy().astype(np.float64).reshape(B, S, H, 64).transpose(0, 2, 1, 3)
    dv = p.transpose(0, 1, 3, 2) @ do
    dp = do @ v.transpose(0, 1, 3, 2)
    Dq = (do * o_used).sum(-1, keepdims=True)
    ds = p * (dp - Dq)
    dq = ds @ k / 8.0
    dk = ds.transpose(0, 1, 3, 2) @ q / 8.0
    ref = np.stack([dq, dk, dv], 2)  # [B,H,3,S,64]
    ref = ref.transpose(0, 3, 2, 1, 4).reshape(B * S, 3 * D)
    got = host(ops.attention_bwd(qkv.cuda(), out, dout.cuda(), lse, B, S, H, causal))
    for i, name in enumerate(("dQ", "dK", "dV")):
        a, b_ = got[:, i * D:(i + 1) * D], ref[:, i * D:(i + 1) * D]
        err = np.abs(a - b_).max()
        assert err <= 3e-2 * max(1.0, np.abs(b_).max()), (name, err, np.abs(b_).max())
        # and in aggregate much tighter than the worst element: bf16 operands, fp32 accumulation
        assert np.sqrt(((a - b_) ** 2).mean()) <= 6e-3 * max(1e-3, np.sqrt((b_ ** 2).mean())), name
