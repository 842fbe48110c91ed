"""Backward of the contrastive loss on an MI355X (first slice) against gradients produced by the
reference's autograd (tests/golden/make_golden_loss_grad.py) and the oracle.  fp32 end to end: tolerances are fp32 round-off."""
import numpy as np
import pytest
import torch
from torch import nn

from oracle import clip_oracle as oc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    from multimodal_amd import build

    build.build()


def host(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def test_reference_gradient_kat_through_user_encoders(golden):
    """The reference's own training-step test: torch Linear encoders (the user's model, ATen) + OUR loss module."""
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature

    z = golden("loss_grad.npz")
    image_encoder, text_encoder = nn.Linear(8, 3), nn.Linear(5, 3)
    with torch.no_grad():
        image_encoder.weight.copy_(torch.from_numpy(z["kat.iw"])); image_encoder.bias.copy_(torch.from_numpy(z["kat.ib"]))
        text_encoder.weight.copy_(torch.from_numpy(z["kat.tw"])); text_encoder.bias.copy_(torch.from_numpy(z["kat.tb"]))
    image_encoder, text_encoder = image_encoder.cuda(), text_encoder.cuda()
    loss_fn = ContrastiveLossWithTemperature().cuda()
    params = list(image_encoder.parameters()) + list(text_encoder.parameters()) + list(loss_fn.parameters())
    opt = torch.optim.SGD(params, lr=1e-4)
    ia = image_encoder(torch.from_numpy(z["kat.image_tensor"]).cuda())
    tb = text_encoder(torch.from_numpy(z["kat.text_tensor"]).cuda())
    ia.retain_grad(); tb.retain_grad()
    loss = loss_fn(ia, tb)
    opt.zero_grad()
    loss.backward()
    assert abs(float(loss) - 3.8848) <= 1e-3
    assert abs(float(image_encoder.weight.grad.mean()) - 0.0979) <= 1e-3
    assert abs(float(text_encoder.bias.grad.mean()) - (-1.8151)) <= 1e-3
    assert abs(float(loss_fn.logit_scale.grad) - 3.6792) <= 1e-3
    assert np.abs(host(ia.grad) - z["kat.grad_emb_a"]).max() <= 2e-5 and np.abs(host(tb.grad) - z["kat.grad_emb_b"]).max() <= 2e-5
    assert np.abs(host(image_encoder.weight.grad) - z["kat.grad_iw"]).max() <= 2e-5
    before = float(loss_fn.logit_scale)
    opt.step()
    assert float(loss_fn.logit_scale) == pytest.approx(before - 1e-4 * 3.6792, abs=1e-6)


@pytest.mark.parametrize("name,kw", [("plain", {}), ("smooth_mask", {"label_smoothing": 0.1}), ("sum", {"reduction": "sum"})])
def test_functional_backward_vs_reference(golden, name, kw):
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import contrastive_loss_with_temperature

    z = golden("loss_grad.npz")
    a = torch.from_numpy(z[f"{name}.a"]).cuda().requires_grad_(True)
    b = torch.from_numpy(z[f"{name}.b"]).cuda().requires_grad_(True)
    s = nn.Parameter(torch.tensor(2.3, device="cuda"))
    mask = torch.from_numpy(z[f"{name}.mask"]).cuda() if f"{name}.mask" in z.files else None
    o = contrastive_loss_with_temperature(a, b, s, mask=mask, cross_entropy_kwargs=kw or None)
    (o.loss * 1.7 + 0.3 * o.loss_a).backward()
    assert abs(float(o.loss) - float(z[f"{name}.loss"])) <= 2e-5 * max(1.0, abs(float(z[f"{name}.loss"])))
    assert np.abs(host(a.grad) - z[f"{name}.grad_a"]).max() <= 2e-5
    assert np.abs(host(b.grad) - z[f"{name}.grad_b"]).max() <= 2e-5
    assert abs(float(s.grad) - float(z[f"{name}.grad_s"])) <= 2e-4 * max(1.0, abs(float(z[f"{name}.grad_s"])))
    assert not o.logits_a.requires_grad  # documented: logits are outputs, not differentiable through the node


def test_two_rank_gradients_simulated_through_the_c_abi(golden):
    """One GPU cannot host two RCCL ranks: both ranks' kernels are run one after the other on the gathered arrays and the
    reduce-scatter (a sum over ranks of the own-block rows) is done on the host — exactly the data flow of _ContrastiveFn.backward."""
    from multimodal_amd import _lib, ops

    z = golden("loss_grad.npz")
    a_all, b_all = torch.from_numpy(z["dist.a_all"]).cuda(), torch.from_numpy(z["dist.b_all"]).cuda()
    W, (WB, E) = 2, a_all.shape
    B = WB // W
    buf = torch.cat([a_all, b_all], 1).contiguous()
    scale = torch.tensor([np.log(1 / 0.07)], dtype=torch.float32, device="cuda")
    g3 = torch.tensor([1.0, 0.0, 0.0], device="cuda")
    per_rank = []
    for r in range(W):
        a, b = a_all[r * B:(r + 1) * B].contiguous(), b_all[r * B:(r + 1) * B].contiguous()
        out3, la, lb = ops.contrastive_fwd(a, b, buf[:, :E], buf[:, E:], 2 * E, scale, label_offset=B * r)
        assert abs(float(out3[0]) - float(z[f"dist.GLOBAL.r{r}.loss"])) <= 2e-5
        ga, gb, g_all, gs = ops.contrastive_bwd(a, b, buf[:, :E], buf[:, E:], 2 * E, scale, la, lb, B * r, None, 0.0, _lib.REDUCE_MEAN,
                                                g3, None, (0, WB))
        ga_l, gb_l, _, _ = ops.contrastive_bwd(a, b, buf[:, :E], buf[:, E:], 2 * E, scale, la, lb, B * r, None, 0.0, _lib.REDUCE_MEAN,
                                               g3, None, (B * r, B), True)
        per_rank.append((host(ga), host(gb), host(g_all), float(gs), host(ga_l), host(gb_l)))
    for r in range(W):
        blk = slice(r * B, (r + 1) * B)
        ga, gb, _, gs, ga_l, gb_l = per_rank[r]
        rs = sum(p[2][blk] for p in per_rank)  # reduce_scatter_tensor(sum)[rank r]
        assert np.abs(ga + rs[:, :E] - z[f"dist.GLOBAL.r{r}.grad_a"]).max() <= 2e-5
        assert np.abs(gb + rs[:, E:] - z[f"dist.GLOBAL.r{r}.grad_b"]).max() <= 2e-5
        assert np.abs(ga_l - z[f"dist.LOCAL.r{r}.grad_a"]).max() <= 2e-5 and np.abs(gb_l - z[f"dist.LOCAL.r{r}.grad_b"]).max() <= 2e-5
        assert np.abs(ga - z[f"dist.NONE.r{r}.grad_a"]).max() <= 2e-5 and np.abs(gb - z[f"dist.NONE.r{r}.grad_b"]).max() <= 2e-5
        assert abs(gs - float(z[f"dist.GLOBAL.r{r}.grad_s"])) <= 1e-4


def test_backward_at_headline_size_vs_oracle_and_bf16_inputs():
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.distributed import BackpropType

    torch.manual_seed(0)
    B, E = 256, 512
    a0 = torch.nn.functional.normalize(torch.randn(B, E), dim=1)
    b0 = torch.nn.functional.normalize(torch.randn(B, E), dim=1)
    loss_fn = ContrastiveLossWithTemperature().cuda()
    ref = oc.contrastive_loss_backward(a0.numpy(), b0.numpy(), np.log(1 / 0.07))
    # no process group here: the reference returns the embeddings themselves from its gather and ignores backprop_type
    # (contrastive_loss_with_temperature.py:31-33), so NONE differentiates through the "gathered" side too
    for bt, extra in ((BackpropType.GLOBAL, True), (BackpropType.LOCAL, True), (BackpropType.NONE, True)):
        a, b = a0.cuda().requires_grad_(True), b0.cuda().requires_grad_(True)
        loss_fn.zero_grad()
        loss_fn(a, b, backprop_type=bt).backward()
        wa = ref["grad_a"] + (ref["grad_a_all"] if extra else 0)
        wb = ref["grad_b"] + (ref["grad_b_all"] if extra else 0)
        assert np.abs(host(a.grad) - wa).max() <= 1e-6 + 1e-4 * np.abs(wa).max()
        assert np.abs(host(b.grad) - wb).max() <= 1e-6 + 1e-4 * np.abs(wb).max()
        assert abs(float(loss_fn.logit_scale.grad) - ref["grad_logit_scale"]) <= 1e-4 * max(1.0, abs(ref["grad_logit_scale"]))
    a = a0.cuda().to(torch.bfloat16).requires_grad_(True)
    b = b0.cuda().to(torch.bfloat16).requires_grad_(True)
    loss_fn(a, b).backward()
    assert a.grad.dtype == torch.bfloat16 and a.grad.shape == (B, E) and torch.isfinite(a.grad.float()).all()
    with torch.no_grad():
        assert loss_fn(a0.cuda(), b0.cuda()).grad_fn is None  # forward-only path untouched
