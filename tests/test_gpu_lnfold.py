"""LayerNorm folded into the GEMMs around it (csrc/gemm.hip "LN fold", include/mmamd.h mmamd_gemm_bf16_res_stats / _lnfold):
kernels against float64 math for every epilogue variant the dispatcher can pick (128x128 direct stores, 256x256 LDS-staged, persistent,
row-range split for long K), and the folded transformer stack against the unfused one."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    from multimodal_amd import build

    build.build()


def f64(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


# (M, N, K): 128-tile kernel / P kernel / persistent kernel / persistent + row-range split (K >= 2048, 2.3 rounds of tiles)
PRODUCER_SHAPES = [(300, 256, 128), (19712, 512, 512), (50432, 768, 768), (50432, 768, 3072), (1000, 1024, 192)]


@pytest.mark.parametrize("M,N,K", PRODUCER_SHAPES)
def test_producer_writes_residual_bf16_copy_and_block_statistics(M, N, K):
    from multimodal_amd import ops

    torch.manual_seed(M + N + K)
    a = (torch.randn(M, K) * 0.5).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(N).cuda()
    x0 = (torch.randn(M, N) * 2 + 0.3).cuda()
    x = x0.clone()
    xh = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    stats = torch.full((M, N // 64, 2), float("nan"), device="cuda")
    ops.gemm_bf16_res_stats(a, w, bias, x, xh, stats)
    plain = ops.gemm_bf16(a, w, bias, residual=x0.clone(), out_dtype=torch.float32)
    assert torch.equal(x, plain)  # the fp32 result is bit-identical to the plain residual GEMM
    assert torch.equal(xh, x.to(torch.bfloat16))  # round-to-nearest-even copy
    blocks = f64(x).reshape(M, N // 64, 64)
    ref = np.stack([blocks.sum(-1), (blocks ** 2).sum(-1)], axis=-1)
    got = f64(stats)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-4


@pytest.mark.parametrize("M,N,K,act", [(300, 256, 128, 0), (19712, 1536, 512, 0), (50432, 3072, 768, 1), (50432, 2304, 768, 0),
                                       (777, 384, 1024, 2)])
def test_consumer_equals_layernorm_then_gemm(M, N, K, act):
    from multimodal_amd import ops

    torch.manual_seed(M + N + K)
    x = (torch.randn(M, K) * 1.5 + 0.2 * torch.randn(M, 1)).cuda()
    gamma, beta = (1 + 0.1 * torch.randn(K)).cuda(), (0.1 * torch.randn(K)).cuda()
    w = (torch.randn(N, K) * 0.05).cuda()
    bias = (0.1 * torch.randn(N)).cuda()
    nslot = K // 64
    xh = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
    stats = torch.empty(M, nslot, 2, device="cuda")
    ops.row_stats(x, xh, stats)
    assert torch.equal(xh, x.to(torch.bfloat16))
    xd = f64(x)
    st = f64(stats)
    assert np.abs(st[:, 0, 0] - xd.sum(1)).max() < 1e-3 and np.abs(st[:, 0, 1] - (xd ** 2).sum(1)).max() < 1e-2 and np.all(st[:, 1:] == 0)
    # spread the sums over the blocks exactly as a producing GEMM would have
    blocks = xd.reshape(M, nslot, 64)
    stats = torch.from_numpy(np.stack([blocks.sum(-1), (blocks ** 2).sum(-1)], -1).astype(np.float32)).cuda()
    wg, c1, c2 = ops.lnfold_pack(w, gamma, beta, bias)
    gw = f64(gamma)[None, :] * f64(w)
    assert np.abs(f64(wg) - gw).max() <= 2 ** -8 * np.abs(gw).max()
    assert np.abs(f64(c1) - f64(wg).sum(1)).max() < 1e-4 and np.abs(f64(c2) - (f64(w) @ f64(beta) + f64(bias))).max() < 1e-5
    out = ops.gemm_bf16_lnfold(xh, wg, c1, c2, stats, 1e-5, act=act)
    # exact math on the operands the kernel multiplies: rstd (xh . Wg - mu c1) + c2
    mu = xd.mean(1, keepdims=True)
    rstd = 1.0 / np.sqrt(xd.var(1, keepdims=True) + 1e-5)
    pre = rstd * (f64(xh) @ f64(wg).T - mu * f64(c1)[None, :]) + f64(c2)[None, :]
    if act == 1:
        pre = pre / (1 + np.exp(-1.702 * pre))
    elif act == 2:
        from math import erf

        pre = 0.5 * pre * (1 + np.vectorize(erf)(pre / np.sqrt(2)))
    got = f64(out)
    assert np.abs(got - pre).max() <= 2 ** -8 * np.abs(pre).max() + 2e-3  # bf16 output rounding
    # and against LayerNorm in exact arithmetic followed by the Linear: the fold's operand rounding differs, the result agrees to bf16 level
    ln = (xd - mu) * rstd * f64(gamma) + f64(beta)
    ref = ln @ f64(w).T + f64(bias)
    if act == 0:
        assert np.abs(got - ref).max() <= 3e-2 * max(1.0, np.abs(ref).max())
