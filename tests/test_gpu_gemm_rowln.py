"""Out-projection + residual + LayerNorm in one launch (multimodal_amd/csrc/gemm_rowln.hip, mmamd_gemm_bf16_residual_ln_grouped) against the two
launches it replaces (mmamd_gemm_bf16 with the fp32 residual, then mmamd_layernorm) and against a float64 restatement of
`x = x + out_proj(a); y = norm2(x)` (nn.TransformerEncoderLayer, norm_first=True: reference models/clip/image_encoder.py:108, text_encoder.py:121)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(M, N, K, seed, dev):
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = (torch.randn(M, K, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * (K ** -0.5)).to(torch.bfloat16).to(dev)
    bias = (torch.randn(N, generator=g) * 0.1).to(dev)
    x = (torch.randn(M, N, generator=g) * 1.5 + 0.3).to(dev)
    gamma = (1.0 + 0.2 * torch.randn(N, generator=g)).to(dev)
    beta = (0.1 * torch.randn(N, generator=g)).to(dev)
    return a, w, bias, x, gamma, beta


def _two_launches(ops, a, w, bias, x, gamma, beta, eps):
    x2 = x.clone()
    ops.gemm_bf16(a, w, bias, residual=x2, out_dtype=torch.float32, out=x2)
    return x2, ops.layernorm(x2, gamma, beta, eps)


def _check(ops, probs, eps=1e-5):
    refs = [_two_launches(ops, *p, eps) for p in probs]
    xs = [p[3].clone() for p in probs]
    ys = [torch.full(p[3].shape, float("nan"), dtype=torch.bfloat16, device=p[3].device) for p in probs]
    ops.gemm_residual_ln_grouped([(p[0], ops.pack_w_ksteps(p[1]), p[2], xs[i], p[4], p[5], eps, ys[i]) for i, p in enumerate(probs)])
    torch.cuda.synchronize()
    for i, p in enumerate(probs):
        x_ref, y_ref = refs[i]
        assert torch.equal(xs[i], x_ref), f"problem {i}: X differs from mmamd_gemm_bf16's by {(xs[i] - x_ref).abs().max().item():.3e}"
        assert not torch.isnan(ys[i].float()).any()
        # Y: the same arithmetic up to the summation order of the row statistics -> at most one bf16 rounding step, on a few elements
        d = (ys[i].float() - y_ref.float()).abs()
        tol = y_ref.float().abs() * 2.0 ** -7 + 1e-6
        assert bool((d <= tol).all()), f"problem {i}: Y off by {d.max().item():.3e}"
        assert (d > 0).float().mean().item() < 0.02
        # float64 restatement of the reference's two statements
        a, w, bias, x, gamma, beta = [t.double() for t in p]
        x64 = x + a @ w.t() + bias
        y64 = torch.nn.functional.layer_norm(x64, (x64.shape[1],), gamma, beta, eps)
        assert (xs[i].double() - x64).abs().max().item() < 2e-4 * max(1.0, x64.abs().max().item())
        assert (ys[i].double() - y64).abs().max().item() < 2.0 ** -7 * max(1.0, y64.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(640, 768, 768), (64, 512, 512), (200, 512, 512), (1, 768, 768), (63, 768, 256), (4160, 768, 3072), (130, 512, 128)])
def test_one_problem(M, N, K):
    from multimodal_amd import ops
    _check(ops, [_problem(M, N, K, 100 + M, torch.device("cuda", 0))])


def test_two_towers_grouped_at_the_headline_shapes_scaled_down():
    from multimodal_amd import ops
    dev = torch.device("cuda", 0)
    _check(ops, [_problem(197 * 96, 768, 768, 1, dev), _problem(77 * 96, 512, 512, 2, dev)])


def test_many_tiles_per_workgroup_and_a_ragged_edge():
    from multimodal_amd import ops
    dev = torch.device("cuda", 0)
    _check(ops, [_problem(64 * 700 + 17, 768, 768, 3, dev), _problem(64 * 300 + 5, 512, 512, 4, dev)])


def test_run_to_run_bit_identity():
    from multimodal_amd import ops
    dev = torch.device("cuda", 0)
    a, w, bias, x, gamma, beta = _problem(64 * 513, 768, 768, 5, dev)
    outs = []
    for _ in range(3):
        x1 = x.clone()
        y1 = torch.empty(x.shape, dtype=torch.bfloat16, device=dev)
        ops.gemm_residual_ln_grouped([(a, ops.pack_w_ksteps(w), bias, x1, gamma, beta, 1e-5, y1)])
        outs.append((x1, y1))
    for x1, y1 in outs[1:]:
        assert torch.equal(x1, outs[0][0]) and torch.equal(y1, outs[0][1])


def test_unsupported_shapes_are_refused():
    from multimodal_amd import ops
    dev = torch.device("cuda", 0)
    assert not ops.gemm_residual_ln_supported(128, 1024, 1024)
    assert ops.gemm_residual_ln_supported(50432, 768, 768) and ops.gemm_residual_ln_supported(19712, 512, 512)
    a, w, bias, x, gamma, beta = _problem(64, 1024, 1024, 6, dev)
    with pytest.raises(ops.MmamdError):
        ops.gemm_residual_ln_grouped([(a, ops.pack_w_ksteps(w), bias, x, gamma, beta, 1e-5, torch.empty(x.shape, dtype=torch.bfloat16, device=dev))])
