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


def test_layernorm_backward_and_colsum():
    from multimodal_amd import ops

    set_rng_seed(3)
    for rows, d, dt in ((37, 128, torch.float32), (1000, 768, torch.bfloat16), (5000, 512, torch.float32), (3, 1024, torch.bfloat16), (4200, 768, torch.float32), (333, 644, torch.bfloat16)):
        x = (torch.randn(rows, d) * 2 + 0.3).requires_grad_(True)
        gamma, beta = (torch.rand(d) + 0.5).requires_grad_(True), torch.randn(d).requires_grad_(True)
        dy = torch.randn(rows, d).to(dt)
        add = torch.randn(rows, d)
        y = torch.nn.functional.layer_norm(x.double(), (d,), gamma.double(), beta.double(), 1e-5)
        y.backward(dy.double())
        dx, dg, db, dxb = ops.layernorm_bwd(x.detach().cuda(), gamma.detach().cuda(), dy.cuda(), 1e-5, add=add.cuda(), want_bf16=True)
        assert torch.equal(dxb, dx.to(torch.bfloat16))
        assert np.abs(host(dx) - (x.grad.double() + add.double()).numpy()).max() <= 2e-5 * max(1.0, float(x.grad.abs().max()))
        assert np.abs(host(dg) - gamma.grad.double().numpy()).max() <= 1e-4 * max(1.0, float(gamma.grad.abs().max()))
        assert np.abs(host(db) - beta.grad.double().numpy()).max() <= 1e-4 * max(1.0, float(beta.grad.abs().max()))
        cs = ops.colsum(dy.cuda())
        assert np.abs(host(cs) - dy.double().sum(0).numpy()).max() <= 1e-4 * max(1.0, float(dy.double().sum(0).abs().max()))
        # the same call with the fused column sums of its own output (bias gradient of the Linear upstream of the residual stream)
        dx2, dg2, db2, dxb2, dxs = ops.layernorm_bwd(x.detach().cuda(), gamma.detach().cuda(), dy.cuda(), 1e-5, add=add.cuda(), want_bf16=True,
                                                    want_colsum=True)
        assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db) and torch.equal(dxb2, dxb)
        want = host(dx).sum(0)
        assert np.abs(host(dxs) - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


def test_layernorm_backward_deferred_reduction_is_bit_identical():
    """r05: the LayerNorm backward calls of a stack park their dgamma / dbeta / column-sum partials and ONE launch reduces all of them
    (mmamd_colsum_stage2_batched) -- the same arithmetic per job as the immediate form."""
    from multimodal_amd import ops

    set_rng_seed(11)
    cases = []
    for rows, d, cs, bf_dy in ((517, 768, True, False), (64, 512, False, True), (3100, 128, True, True), (9, 2048, True, False), (4300, 768, True, True)):
        dy = torch.randn(rows, d).cuda()
        cases.append((torch.randn(rows, d).cuda(), torch.randn(d).cuda(), dy.to(torch.bfloat16) if bf_dy else dy, torch.randn(rows, d).cuda(), cs))
    want = [ops.layernorm_bwd(x, g, dy, 1e-5, add=add, want_bf16=True, want_colsum=cs) for x, g, dy, add, cs in cases]
    pending = []
    got = [ops.layernorm_bwd(x, g, dy, 1e-5, add=add, want_bf16=True, want_colsum=cs, defer=pending) for x, g, dy, add, cs in cases]
    assert len(pending) == len(cases)
    ops.colsum_flush(pending)  # one launch for the four jobs
    assert pending == []
    for w, o in zip(want, got):
        assert len(w) == len(o)
        for a, b in zip(w, o):
            assert torch.equal(a, b)


def test_activation_transpose_normalize_scatter_kernels():
    from multimodal_amd import ops

    set_rng_seed(4)
    u = (torch.randn(300, 256) * 2).to(torch.bfloat16)
    dg = torch.randn(300, 256).to(torch.bfloat16)
    for act, fn in ((ops.ACT_QUICKGELU, lambda t: t * torch.sigmoid(1.702 * t)), (ops.ACT_GELU_ERF, torch.nn.functional.gelu)):
        ud = u.double().requires_grad_(True)
        y = fn(ud)
        y.backward(dg.double())
        assert np.abs(host(ops.act_fwd(u.cuda(), act)) - y.detach().numpy()).max() <= 2 ** -7 * max(1.0, float(y.abs().max()))
        assert np.abs(host(ops.act_bwd(u.cuda(), dg.cuda(), act)) - ud.grad.numpy()).max() <= 2 ** -6 * max(1.0, float(ud.grad.abs().max()))
    for rows, cols, dt in ((197, 128, torch.float32), (1000, 768, torch.bfloat16), (5, 3, torch.float32)):
        src = torch.randn(rows, cols + 8).to(dt)
        t, cs = ops.transpose_to_bf16(src.cuda()[:, :cols], pad_to=64, with_colsum=True)
        ld = (rows + 63) // 64 * 64
        assert t.shape == (cols, ld)
        assert torch.equal(t[:, :rows].cpu(), src[:, :cols].to(torch.bfloat16).t()) and not t[:, rows:].any()
        want = src[:, :cols].to(torch.bfloat16).double().sum(0).numpy()
        assert np.abs(host(cs) - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
        assert torch.equal(ops.transpose_to_bf16(src.cuda()[:, :cols], pad_to=128)[:, :rows], t[:, :rows])
    x = torch.randn(50, 64, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(50, 64, dtype=torch.float64)
    torch.nn.functional.normalize(x, dim=1).backward(dy)
    got = ops.l2_normalize_bwd(x.detach().float().cuda(), dy.float().cuda())
    assert np.abs(host(got) - x.grad.numpy()).max() <= 1e-5
    dst = torch.zeros(20, 32)
    idx = torch.randint(0, 20, (500,))
    src = torch.randn(500, 32)
    d = dst.cuda()
    ops.scatter_add_rows_(d, idx.cuda(), src.cuda())
    assert np.abs(host(d) - dst.double().index_add_(0, idx, src.double()).numpy()).max() <= 1e-4
    X, Y = torch.randn(7, 33), torch.randn(5, 33)
    C = ops.f32_gemm_strided(X.cuda(), 33, 1, Y.cuda(), 33, 1, 7, 5, 33)
    assert np.abs(host(C) - (X.double() @ Y.double().t()).numpy()).max() <= 1e-5
    C2 = ops.f32_gemm_strided(X.cuda(), 1, 33, Y.cuda(), 1, 33, 33, 33, 5)  # X^T Y over the first 5 rows
    assert np.abs(host(C2) - (X[:5].double().t() @ Y.double()).numpy()).max() <= 1e-5
    # split-K weight-gradient GEMM: long contraction, few output tiles
    for (M, N, K) in ((768, 3072, 50432), (128, 256, 512), (2304, 768, 6400)):
        a = torch.randn(M, K).to(torch.bfloat16)
        w = (torch.randn(N, K) * 0.1).to(torch.bfloat16)
        got = host(ops.gemm_bf16_splitk(a.cuda(), w.cuda()))
        ref = host(a.cuda().float() @ w.cuda().float().t())  # fp32 reference of the same bf16 operands (ATen, test only)
        assert np.abs(got - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max()), (M, N, K)


def test_gemm_epilogue_multiplies_by_activation_gradient():
    """dgrad of the MLP with the activation's backward in the epilogue: (dY W) * act'(u) for QuickGELU and erf-GELU, every kernel path."""
    from multimodal_amd import ops

    set_rng_seed(9)
    # (30011 x 776: a ragged last row tile AND a ragged last column tile on the persistent kernel -- the buffer-descriptor range checks of its R / C accesses)
    for (M, N, K) in ((300, 256, 128), (50432, 3072, 768), (5000, 768, 256), (30011, 776, 128)):
        a = torch.randn(M, K).to(torch.bfloat16)
        w = (torch.randn(N, K) * 0.05).to(torch.bfloat16)
        u = (torch.randn(M, N) * 2).to(torch.bfloat16)
        base = (a.cuda().float() @ w.cuda().float().t()).double().cpu()  # fp32 reference of the same bf16 operands (test only)
        for code, fn in ((ops.ACT_MUL_QUICKGELU_GRAD, lambda t: t * torch.sigmoid(1.702 * t)), (ops.ACT_MUL_GELU_GRAD, torch.nn.functional.gelu)):
            ud = u.double().requires_grad_(True)
            fn(ud).sum().backward()
            ref = (base * ud.grad).numpy()
            before = ops.launch_count("gemm_bf16_pp")
            got = host(ops.gemm_bf16(a.cuda(), w.cuda(), None, act=code, residual=u.cuda()))
            if M >= 30000:  # the large shapes take the persistent kernel (its descriptor epilogue)
                assert ops.launch_count("gemm_bf16_pp") == before + 1, (M, N, K)
            assert np.abs(got - ref).max() <= 2 ** -6 * max(1.0, np.abs(ref).max()), (M, N, K, code)
    with pytest.raises(ops.MmamdError):
        ops.gemm_bf16(a.cuda(), w.cuda(), None, act=ops.ACT_MUL_GELU_GRAD)  # needs the saved pre-activation


def test_clip_training_step_gradients_vs_reference_autograd(golden):
    """Every parameter gradient of a CLIP training step (two towers + contrastive loss) against the reference's torch autograd."""
    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from tests._util import fixture_sd

    z, zg = golden("midsize.npz"), golden("clip_grad.npz")
    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=2, patch_size=16, image_size=64, width=128)
    txt = CLIPTextEncoder(embedding_dim=64, context_length=77, vocab_size=1000, width=128, dim_feedforward=256, heads=2, layers=2)
    clip = CLIP(vit, txt)
    clip.load_state_dict({k: torch.from_numpy(v) for k, v in fixture_sd(z).items()}, strict=True)
    clip = clip.cuda().train()
    loss_fn = ContrastiveLossWithTemperature().cuda()
    out = clip(torch.from_numpy(z["images"]).cuda(), torch.from_numpy(z["ids"]).cuda())
    assert out.embeddings_a.requires_grad and out.embeddings_a.grad_fn is not None
    out.embeddings_a.retain_grad(); out.embeddings_b.retain_grad()
    loss = loss_fn(out.embeddings_a, out.embeddings_b)
    loss.backward()
    print("train-mode loss", float(loss), "reference", float(zg["loss"]))
    assert abs(float(loss) - float(zg["loss"])) <= 1e-2  # the pre-activation u is kept in bf16 for the backward: one more rounding than eval
    report = {}
    worst = ("", 0.0)
    for k, p in list(clip.named_parameters()) + [("logit_scale", loss_fn.logit_scale)]:
        assert p.grad is not None, k
        ref = zg["g." + k].astype(np.float64)
        got = host(p.grad)
        assert got.shape == ref.shape, k
        scale = max(np.abs(ref).max(), 1e-6)
        rel = np.abs(got - ref).max() / scale
        rms = np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-9)
        report[k] = (rel, rms)
        if rel > worst[1]:
            worst = (k, rel)
    print("clip grad parity: worst max-rel", worst, " median max-rel", float(np.median([v[0] for v in report.values()])),
          " median rms-rel", float(np.median([v[1] for v in report.values()])))
    for k, (rel, rms) in report.items():
        assert rel <= 6e-2 and rms <= 3e-2, (k, rel, rms)  # bf16 MFMA operands through 2 x 2 layers, fp32 accumulation
    assert np.abs(host(out.embeddings_a.grad) - zg["grad_emb_a"]).max() <= 2e-2 * np.abs(zg["grad_emb_a"]).max()
    # an optimizer step runs on the ordinary nn.Parameters
    opt = torch.optim.SGD(list(clip.parameters()) + list(loss_fn.parameters()), lr=1e-3)
    before = clip.encoder_a.projection.detach().clone()
    opt.step()
    assert not torch.equal(before, clip.encoder_a.projection.detach())
    with torch.no_grad():
        clip.eval()
        out2 = clip(torch.from_numpy(z["images"]).cuda(), torch.from_numpy(z["ids"]).cuda())
    assert out2.embeddings_a.grad_fn is None


def test_training_step_under_ddp_single_rank_rccl():
    """DistributedDataParallel over RCCL wraps the drop-in CLIP (parameters are ordinary nn.Parameters, gradients ordinary tensors):
    same gradients as without DDP; the loss's GLOBAL backprop goes through its reduce-scatter-free single-rank path."""
    import os

    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        set_rng_seed(5)
        vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=1, patch_size=16, image_size=32, width=128)
        txt = CLIPTextEncoder(embedding_dim=64, context_length=77, vocab_size=300, width=128, dim_feedforward=256, heads=2, layers=1)
        clip = CLIP(vit, txt).cuda().train()
        loss_fn = ContrastiveLossWithTemperature().cuda()
        images, ids = clip_batch(8, image_size=32, vocab_size=300)
        images, ids = images.cuda(), ids.cuda()

        def grads(model):
            for p in list(model.parameters()) + list(loss_fn.parameters()):
                p.grad = None
            out = model(images, ids)
            loss_fn(out.embeddings_a, out.embeddings_b).backward()
            return {k: p.grad.detach().clone() for k, p in clip.named_parameters()}, loss_fn.logit_scale.grad.clone()

        g_plain, s_plain = grads(clip)
        ddp = DDP(clip, device_ids=[0])
        g_ddp, s_ddp = grads(ddp)
        for k in g_plain:  # all-reduce over one rank = identity; the embedding-table gradient uses fp32 atomics (order varies)
            assert torch.allclose(g_plain[k], g_ddp[k], rtol=1e-4, atol=1e-6 * float(g_plain[k].abs().max() + 1e-12)), k
        assert torch.allclose(s_plain, s_ddp, rtol=1e-5)
    finally:
        if created:
            dist.destroy_process_group()


def test_flava_training_step_gradients_vs_reference_autograd(golden):
    """FLAVA dual encoder + multimodal encoder + global contrastive loss in train mode: every reached parameter's gradient against
    the reference's torch autograd (padded text -> key masks in the attention backward; shared encoders used twice)."""
    from multimodal_amd.models.flava.model import flava_model
    from multimodal_amd.modules.losses.flava import FLAVAGlobalContrastiveLoss
    from tests._util import fixture_sd
    from tests.golden.make_golden_flava_grad import SMALL_KW

    z, zg = golden("flava_small.npz"), golden("flava_grad.npz")
    model = flava_model(**SMALL_KW)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fixture_sd(z).items()}, strict=True)
    model = model.cuda().train()
    loss_mod = FLAVAGlobalContrastiveLoss().cuda()
    image, text, text_masked = (torch.from_numpy(z[k]).cuda() for k in ("image", "text", "text_masked"))
    out = model(image, text, text_masked=text_masked)
    # training mode hands out what the reference does: every hidden state (attached) and the attention probabilities (values)
    assert len(out.image.attentions) == 2 and len(out.image.hidden_states) == 3 and out.image.hidden_states[1].requires_grad
    itc = loss_mod(out.projected_image_embeddings, out.projected_text_embeddings, torch.ones(image.shape[0], dtype=torch.bool, device="cuda")).loss
    probe = torch.linspace(-1.0, 1.0, 128, device="cuda")
    mm_term = (out.multimodal_masked.last_hidden_state[:, 0] * probe).sum(-1).mean()
    (itc + mm_term).backward()
    assert abs(float(itc) - float(zg["itc"])) <= 1e-2 and abs(float(mm_term) - float(zg["mm_term"])) <= 6e-2  # a 128-term sum of LN outputs
    no_grad = {str(k) for k in zg["no_grad_keys"]}
    report, worst = {}, ("", 0.0)
    for k, p in list(model.named_parameters()) + [("logit_scale", loss_mod.logit_scale)]:
        if k in no_grad:
            assert p.grad is None or not p.grad.any(), k
            continue
        assert p.grad is not None, k
        ref = zg["g." + k].astype(np.float64)
        got = host(p.grad)
        assert got.shape == ref.shape, k
        if np.abs(ref).max() < 1e-6:
            # mathematically zero gradients (a key bias shifts every score of a query by the same amount: softmax-invariant): the
            # reference holds fp32 round-off there, this path bf16 round-off of the dK rows it sums — both must be ~0
            assert k.endswith("attention.key.bias") and np.abs(got).max() <= 2e-3, (k, np.abs(got).max())
            continue
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        rms = np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-12)
        report[k] = (rel, rms)
        if rel > worst[1]:
            worst = (k, rel)
    print("flava grad parity: worst max-rel", worst, " median max-rel", float(np.median([v[0] for v in report.values()])),
          " median rms-rel", float(np.median([v[1] for v in report.values()])), " tensors", len(report))
    for k, (rel, rms) in report.items():
        assert rel <= 8e-2 and rms <= 4e-2, (k, rel, rms)


def test_cross_entropy_backward_kernel():
    from multimodal_amd import ops

    set_rng_seed(21)
    for (N, V, pad, dt) in ((300, 30522, 64, torch.bfloat16), (5, 2, 1, torch.float32), (64, 200, 64, torch.bfloat16), (40, 49408, 64, torch.bfloat16),
                            (33, 1024, 4, torch.float32)):  # (the last two take the vectorised one-read / 4-columns-per-thread form)
        logits = (torch.randn(N, V) * 3).requires_grad_(True)
        lab = torch.randint(0, V, (N,))
        lab[torch.rand(N) < 0.3] = -1
        lab[0] = 1
        loss = torch.nn.functional.cross_entropy(logits.double(), lab, ignore_index=-1)
        (loss * 1.7).backward()
        g = ops.cross_entropy_bwd(logits.detach().cuda(), lab.cuda(), -1, torch.tensor([1.7], device="cuda"), out_dtype=dt, pad_cols_to=pad)
        assert g.shape == (N, (V + pad - 1) // pad * pad) and not g[:, V:].any()
        ref = logits.grad.double().numpy()
        tol = 1e-6 if dt == torch.float32 else 2 ** -8 * np.abs(ref).max()
        assert np.abs(host(g[:, :V]) - ref).max() <= tol + 1e-7


def test_cross_entropy_with_masked_minus_inf_logits():
    """Rows whose FIRST columns are -inf (a masked vocabulary prefix): torch returns a finite loss and gradient; the one-pass running
    (max, sum) must not form -inf - -inf (the scalar path, V % 4 != 0 such as FLAVA's 30522, and the vectorised path)."""
    from multimodal_amd import ops

    set_rng_seed(22)
    for (N, V, pad, dt) in ((7, 30522, 64, torch.float32), (9, 49408, 64, torch.float32), (5, 1000, 4, torch.bfloat16), (4, 301, 1, torch.float32)):
        logits = torch.randn(N, V) * 3
        logits[:, :300] = -float("inf")          # every thread's first element / first 16-byte chunk is -inf
        logits[1, 300:V - 2] = -float("inf")     # only the tail of a row is finite
        lab = torch.randint(300, V, (N,))
        lab[1] = V - 1
        if V > 302:
            logits[2, ::2] = -float("inf")
            lab[2] = 301
        logits.requires_grad_(True)
        loss = torch.nn.functional.cross_entropy(logits.double(), lab)
        assert torch.isfinite(loss)
        (loss * 1.3).backward()
        got = ops.cross_entropy(logits.detach().cuda(), lab.cuda(), -1)
        assert abs(float(got) - float(loss)) <= 2e-6 * max(1.0, abs(float(loss))), (V, float(got), float(loss))
        g = ops.cross_entropy_bwd(logits.detach().cuda(), lab.cuda(), -1, torch.tensor([1.3], device="cuda"), out_dtype=dt, pad_cols_to=pad)
        ref = logits.grad.double().numpy()
        assert np.isfinite(host(g)).all()
        tol = 1e-6 if dt == torch.float32 else 2 ** -8 * np.abs(ref).max()
        assert np.abs(host(g[:, :V]) - ref).max() <= tol + 1e-7


@pytest.mark.parametrize("batched_train", [True, False])
def test_flava_full_pretraining_step_gradients_vs_reference_autograd(golden, batched_train):
    """The whole FLAVA pre-training objective (ITM + MMM text/image heads + global contrastive; patch mask, padded text, ITM row
    filter) in train mode: every parameter gradient of the model AND of the loss heads against the reference's torch autograd -- with the
    unmasked and the masked pass of each tower as ONE 2B pass (schedule.flava_batched_train, the default: every parameter receives one
    gradient) and as the reference's two passes (autograd adds the two)."""
    from multimodal_amd.schedule import get_schedule, set_schedule

    prev = get_schedule().flava_batched_train
    set_schedule(flava_batched_train=batched_train)
    try:
        _flava_pretraining_grads_vs_reference(golden)
    finally:
        set_schedule(flava_batched_train=prev)


def _flava_pretraining_grads_vs_reference(golden):
    from multimodal_amd.models.flava.model import flava_model
    from multimodal_amd.modules.losses.flava import FLAVAPretrainingLoss
    from tests._util import fixture_sd
    from tests.golden.make_golden_flava_grad import SMALL_KW

    z, zl, zg = golden("flava_small.npz"), golden("flava_pretrain_small.npz"), golden("flava_pretrain_grad.npz")
    model = flava_model(**SMALL_KW)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fixture_sd(z).items()}, strict=True)
    loss = FLAVAPretrainingLoss(hidden_size=128, text_vocab_size=200, image_vocab_size=64)
    loss.load_state_dict({k: torch.from_numpy(v) for k, v in fixture_sd(zl).items()}, strict=True)
    model, loss = model.cuda().train(), loss.cuda().train()
    T = lambda a: torch.from_numpy(np.asarray(a)).cuda()
    out = model(T(z["image"]), T(z["text"]), image_patches_mask=T(z["patches_mask"]), text_masked=T(z["text_masked"]))
    lo = loss(image_sequence=out.image.last_hidden_state, text_sequence=out.text.last_hidden_state,
              image_masked_sequence=out.image_masked.last_hidden_state, text_masked_sequence=out.text_masked.last_hidden_state,
              multimodal_masked_sequence=out.multimodal_masked.last_hidden_state, itm_labels=T(zl["itm_labels"]),
              mim_labels=T(zl["mim_labels"]), mlm_labels=T(zl["mlm_labels"]),
              projected_image_embeddings=out.projected_image_embeddings, projected_text_embeddings=out.projected_text_embeddings)
    names = ("itm_loss", "mmm_text_loss", "mmm_image_loss", "global_contrastive_loss")
    total = sum(getattr(lo.losses, n) for n in names)
    total.backward()
    for n in names:
        assert abs(float(getattr(lo.losses, n)) - float(zg[n])) <= 2e-2, (n, float(getattr(lo.losses, n)), float(zg[n]))
    no_grad = {str(k) for k in zg["no_grad_keys"]}
    report, worst = {}, ("", 0.0)
    params = [("model." + k, p) for k, p in model.named_parameters()] + [("loss." + k, p) for k, p in loss.named_parameters()]
    for k, p in params:
        if k in no_grad:
            assert p.grad is None or not p.grad.any(), k
            continue
        assert p.grad is not None, k
        ref = zg["g." + k].astype(np.float64)
        got = host(p.grad)
        assert got.shape == ref.shape, k
        if np.abs(ref).max() < 1e-6:
            assert np.abs(got).max() <= 2e-3, (k, np.abs(got).max())  # mathematically zero (key biases), round-off on both sides
            continue
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        rms = np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-12)
        report[k] = (rel, rms)
        if rel > worst[1]:
            worst = (k, rel)
    print("flava pre-training grad parity: worst max-rel", worst, " median max-rel", float(np.median([v[0] for v in report.values()])),
          " median rms-rel", float(np.median([v[1] for v in report.values()])), " tensors", len(report))
    for k, (rel, rms) in report.items():
        assert rel <= 8e-2 and rms <= 4e-2, (k, rel, rms)


@pytest.mark.parametrize("B,Sq,Sk,H,hd,causal,kmask,fmask,shared", [
    (2, 12, 12, 2, 64, True, False, False, False), (3, 76, 256, 12, 64, False, False, False, False),
    (2, 257, 256, 8, 96, False, False, False, True), (2, 1, 256, 8, 96, False, False, False, True),
    (2, 77, 77, 3, 64, False, False, True, False), (2, 77, 77, 2, 96, True, True, False, False),
    (1, 140, 288, 1, 64, False, True, True, False), (2, 50, 49, 8, 64, False, False, False, True), (2, 130, 130, 2, 96, True, False, False, False)])
def test_attention_x_backward(B, Sq, Sk, H, hd, causal, kmask, fmask, shared):
    """Backward of the general attention (cross-attention, 96-wide heads, masks, batch-shared queries) vs float64 autograd math."""
    from multimodal_amd import ops

    set_rng_seed(Sq * 5 + Sk + hd)
    D = H * hd
    qrows = Sq if shared else B * Sq
    wide = torch.randn(qrows, D + 64).to(torch.bfloat16)
    kv = torch.randn(B * Sk, 2 * D).to(torch.bfloat16)
    dout = torch.randn(B * Sq, D).to(torch.bfloat16)
    km = fm = None
    if kmask:
        km = (torch.rand(B, Sk) > 0.3).to(torch.uint8)
        km[:, 0] = 1
    if fmask:
        fm = (torch.rand(B, Sq, Sk) > 0.4).to(torch.uint8)
        fm[:, :, 0] = 1
    wg, kvg = wide.cuda(), kv.cuda()
    mask = ops.AttnMask(causal=causal, key_mask=km.cuda() if km is not None else None, full=fm.cuda() if fm is not None else None)
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device="cuda")
    out, _ = ops.attention_x_fwd(wg[:, 64:], kvg[:, :D], kvg[:, D:], B, Sq, Sk, H, hd, mask, shared_q=shared, lse=lse)
    dq, dkv = ops.attention_x_bwd(wg[:, 64:], kvg[:, :D], kvg[:, D:], out, dout.cuda(), lse, B, Sq, Sk, H, hd, mask, shared_q=shared)
    q = wide[:, 64:].float().numpy().astype(np.float64)
    q = np.broadcast_to(q.reshape(1, Sq, H, hd), (B, Sq, H, hd)) if shared else q.reshape(B, Sq, H, hd)
    q = q.transpose(0, 2, 1, 3)
    k = kv[:, :D].float().numpy().astype(np.float64).reshape(B, Sk, H, hd).transpose(0, 2, 1, 3)
    v = kv[:, D:].float().numpy().astype(np.float64).reshape(B, Sk, H, hd).transpose(0, 2, 1, 3)
    s = q @ k.transpose(0, 1, 3, 2) / np.sqrt(hd)
    allow = np.ones((B, 1, Sq, Sk), dtype=bool)
    if causal:
        allow = allow & np.tril(np.ones((Sq, Sk), dtype=bool))
    if km is not None:
        allow = allow & km.numpy().astype(bool)[:, None, None, :]
    if fm is not None:
        allow = allow & fm.numpy().astype(bool)[:, None]
    s = np.where(allow, s, -np.inf)
    p = np.exp(s - s.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    o_used = host(out).reshape(B, Sq, H, hd).transpose(0, 2, 1, 3)
    do = dout.float().numpy().astype(np.float64).reshape(B, Sq, H, hd).transpose(0, 2, 1, 3)
    rdv = p.transpose(0, 1, 3, 2) @ do
    dp = do @ v.transpose(0, 1, 3, 2)
    ds = p * (dp - (do * o_used).sum(-1, keepdims=True))
    rdq = (ds @ k / np.sqrt(hd)).transpose(0, 2, 1, 3).reshape(B * Sq, D)
    rdk = (ds.transpose(0, 1, 3, 2) @ q / np.sqrt(hd)).transpose(0, 2, 1, 3).reshape(B * Sk, D)
    rdv = rdv.transpose(0, 2, 1, 3).reshape(B * Sk, D)
    for name, a, b_ in (("dQ", host(dq), rdq), ("dK", host(dkv[:, :D]), rdk), ("dV", host(dkv[:, D:]), rdv)):
        err = np.abs(a - b_).max()
        assert err <= 3e-2 * max(1.0, np.abs(b_).max()), (name, err, np.abs(b_).max())
        assert np.sqrt(((a - b_) ** 2).mean()) <= 6e-3 * max(1e-3, np.sqrt((b_ ** 2).mean())), name


@pytest.mark.parametrize("prefix,kw_name,seed_v,fixture", [("s64.", "SMALL", 51, "coca_small.npz"), ("p96.", "POOL96", 53, "coca_pool96.npz")])
def test_coca_training_step_gradients_vs_reference_autograd(golden, prefix, kw_name, seed_v, fixture):
    """CoCaForPretraining (ViT without CLS, attention pooler with batch-shared queries, causal text decoder with the padding-aware
    mask, cross-attention multimodal decoder, vocabulary projection, contrastive + captioning losses) in train mode: parameter
    gradients against the reference's torch autograd (64-wide heads: every tensor; 96-wide pooler heads: the pooler path)."""
    from multimodal_amd.models.coca.coca_model import coca_vit, CoCaForPretraining
    from tests.golden import make_golden_coca as mg
    from tests.golden.make_golden import seed

    z, zg = golden(fixture), golden("coca_grad.npz")
    seed(seed_v)
    model = coca_vit(**getattr(mg, kw_name), cascaded_pooler=False)
    mg.randomize(model, torch.Generator().manual_seed(seed_v + 1))
    pre = CoCaForPretraining(model).cuda().train()
    losses = pre(torch.from_numpy(z["par.images"]).cuda(), torch.from_numpy(z["par.texts"]).cuda())
    (losses["contrastive"] + losses["captioning"]).backward()
    assert abs(float(losses["contrastive"]) - float(zg[prefix + "contrastive"])) <= 2e-2
    assert abs(float(losses["captioning"]) - float(zg[prefix + "captioning"])) <= 2e-2
    report, worst, checked = {}, ("", 0.0), 0
    for k, p in pre.named_parameters():
        assert p.grad is not None, k
        key = prefix + "g." + k
        if key not in zg.files:
            continue
        ref = zg[key].astype(np.float64)
        got = host(p.grad)
        assert got.shape == ref.shape, k
        checked += 1
        if np.abs(ref).max() < 1e-6:
            assert np.abs(got).max() <= 2e-3, (k, np.abs(got).max())  # mathematically zero (key biases)
            continue
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        rms = np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-12)
        report[k] = (rel, rms)
        if rel > worst[1]:
            worst = (k, rel)
    print(f"coca {prefix} grad parity: worst max-rel", worst, " median max-rel", float(np.median([v[0] for v in report.values()])),
          " median rms-rel", float(np.median([v[1] for v in report.values()])), " tensors", checked)
    assert checked >= (100 if prefix == "s64." else 10)
    for k, (rel, rms) in report.items():
        # the softmax path of the multimodal decoder's cross-attention (q / k projections and the LayerNorm in front) sees 6 keys with
        # near-uniform probabilities in this small model: dS = P (dP - D) is a difference of close numbers, so the bf16 operand
        # rounding shows up as ~5 % there (measured 5.5 % rms worst); everything else is at the 1 % level (median 1.1 %)
        soft = "cross_attention.q_proj" in k or "cross_attention.k_proj" in k or "cross_attention_layernorm" in k
        assert rel <= (1e-1 if soft else 8e-2) and rms <= (8e-2 if soft else 4e-2), (k, rel, rms)


def test_training_loop_overfits_a_fixed_batch():
    """End-to-end trainability: AdamW on the drop-in CLIP + loss drives the contrastive loss of a fixed batch towards zero."""
    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    set_rng_seed(123)
    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=2, patch_size=16, image_size=32, width=128)
    txt = CLIPTextEncoder(embedding_dim=64, context_length=77, vocab_size=300, width=128, dim_feedforward=256, heads=2, layers=2)
    clip = CLIP(vit, txt).cuda().train()
    loss_fn = ContrastiveLossWithTemperature().cuda()
    opt = torch.optim.AdamW(list(clip.parameters()) + list(loss_fn.parameters()), lr=1e-3, weight_decay=0.0)
    images, ids = clip_batch(16, image_size=32, vocab_size=300)
    images, ids = images.cuda(), ids.cuda()
    history = []
    for _ in range(40):
        opt.zero_grad(set_to_none=True)
        out = clip(images, ids)
        loss = loss_fn(out.embeddings_a, out.embeddings_b)
        loss.backward()
        opt.step()
        history.append(float(loss.detach()))
    assert all(np.isfinite(history)) and history[0] > 2.0  # ~ln(16) = 2.77 at initialisation
    assert history[-1] < 0.25 * history[0], history[::8]
    clip.eval()
    with torch.no_grad():
        out = clip(images, ids)
    assert (out.embeddings_a @ out.embeddings_b.t()).argmax(1).eq(torch.arange(16, device="cuda")).float().mean() >= 0.9


def test_gemm_dual_output_matches_gemm_then_activation():
    """Training forward of linear1: one GEMM writes the pre-activation and its activation, on every kernel path (128-tiles, 256-tiles
    with the row-range split, persistent).  The pre-activation must equal the plain GEMM bit for bit; the activation is that of the
    STORED bf16 pre-activation (what mmamd_act_fwd computes from it: same value up to one bf16 rounding tie — the epilogue uses the
    rational erf, the standalone kernel libm's)."""
    from multimodal_amd import ops

    set_rng_seed(21)
    for (M, N, K) in ((300, 256, 128), (50432, 768, 128), (19712, 2048, 512), (50432, 3072, 768), (30011, 776, 128)):
        a = torch.randn(M, K).to(torch.bfloat16).cuda()
        w = (torch.randn(N, K) * 0.05).to(torch.bfloat16).cuda()
        b = torch.randn(N).cuda()
        for act in (ops.ACT_QUICKGELU, ops.ACT_GELU_ERF):
            u, g = ops.gemm_bf16_dual(a, w, b, act)
            u_ref = ops.gemm_bf16(a, w, b)
            assert torch.equal(u, u_ref), (M, N, K, act)
            g_ref = ops.act_fwd(u_ref, act).float()
            diff = (g.float() - g_ref).abs()
            assert bool((diff <= 2.0 ** -7 * g_ref.abs() + 1e-6).all()), (M, N, K, act)   # one bf16 ulp (+ the rational erf's 1.5e-7 |u| in the far negative tail)
            assert float((diff > 1e-6).float().mean()) < 1e-3, (M, N, K, act)
    with pytest.raises(ops.MmamdError):
        ops.gemm_bf16_dual(a, w, b, ops.ACT_NONE)


def test_weight_gradient_gemm_from_row_major_operands():
    """dW = dY^T X straight from the row-major bf16 dY [T, M] and X [T, N] (LDS transpose reads) == the fp32 product of the same
    operands; shapes with partial tiles in both output dimensions and every split count."""
    from multimodal_amd import ops

    set_rng_seed(33)
    for (T, M, N) in ((256, 64, 64), (50432, 768, 3072), (6400, 2304, 768), (1280, 200, 520), (19712, 512, 2048), (128, 8, 8)):
        y = (torch.randn(T, M) * 0.1).to(torch.bfloat16).cuda()
        x = torch.randn(T, N).to(torch.bfloat16).cuda()
        got = host(ops.gemm_bf16_tn_splitk(y, x))
        ref = host(y.float().t() @ x.float())  # fp32 reference of the same bf16 operands (ATen, test only)
        assert np.abs(got - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max()), (T, M, N)
    with pytest.raises(ops.MmamdError):
        ops.gemm_bf16_tn_splitk(y[:64], x[:64])  # token count must be a multiple of 128


def test_weight_gradient_gemm_with_fused_bias_gradient():
    """mmamd_gemm_bf16_tn_splitk_colsum (r05): dW bit-identical to the plain TN split-K GEMM, and db = the column sums of dY summed in fp32
    from the same bf16 values -- against float64 sums, and run-to-run bit-identical (fixed summation order: no atomics).  Shapes: every
    split count, partial tiles in M (columns past the matrix are clamped duplicates that must not leak into db), M < 256, one split."""
    from multimodal_amd import ops

    set_rng_seed(34)
    for (T, M, N) in ((256, 64, 64), (50432, 768, 3072), (6400, 2304, 768), (1280, 200, 520), (19712, 512, 2048), (128, 8, 8), (3840, 3072, 768),
                      (128, 1544, 128)):
        y = (torch.randn(T, M) * 0.1 + 0.01).to(torch.bfloat16).cuda()
        x = torch.randn(T, N).to(torch.bfloat16).cuda()
        dW0 = ops.gemm_bf16_tn_splitk(y, x)
        dW, db = ops.gemm_bf16_tn_splitk(y, x, want_colsum=True)
        assert torch.equal(dW, dW0), (T, M, N)
        ref = y.double().sum(0)
        scale = float(y.double().abs().sum(0).max())
        assert float((db.double() - ref).abs().max()) <= 2e-6 * scale, (T, M, N, float((db.double() - ref).abs().max()), scale)
        dW2, db2 = ops.gemm_bf16_tn_splitk(y, x, want_colsum=True)
        assert torch.equal(db, db2) and torch.equal(dW, dW2), (T, M, N)
    # the autograd helper takes it for full-size batches and agrees with the two-pass form
    from multimodal_amd import _autograd

    y = (torch.randn(1024, 384) * 0.1).to(torch.bfloat16).cuda()
    x = torch.randn(1024, 256).to(torch.bfloat16).cuda()
    dW, db = _autograd.wgrad(y, x, bias=True)
    _autograd._FUSED_BIAS_GRAD = False
    try:
        dW_, db_ = _autograd.wgrad(y, x, bias=True)
    finally:
        _autograd._FUSED_BIAS_GRAD = True
    assert torch.equal(dW, dW_) and float((db - db_).abs().max()) <= 1e-5 * float(db_.abs().max() + 1.0)


def test_small_fp32_linear_with_relu_backward():
    """Classifier-head Linear(+ReLU) in exact fp32 (mmamd_rows_linear_f32 / mmamd_relu_bwd / strided fp32 GEMMs) vs float64 autograd."""
    from multimodal_amd._autograd import SmallLinearF32Fn

    set_rng_seed(17)
    x = torch.randn(9, 40)
    w, b = torch.randn(24, 40) * 0.3, torch.randn(24)
    dy = torch.randn(9, 24)
    for relu in (False, True):
        xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
        yd = xd @ wd.t() + bd
        if relu:
            yd = torch.relu(yd)
        yd.backward(dy.double())
        xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
        y = SmallLinearF32Fn.apply(xg, wg, bg, relu)
        y.backward(dy.cuda())
        assert np.abs(host(y) - yd.detach().numpy()).max() <= 1e-5
        for got, ref in ((xg.grad, xd.grad), (wg.grad, wd.grad), (bg.grad, bd.grad)):
            assert np.abs(host(got) - ref.numpy()).max() <= 1e-5 * max(1.0, float(ref.abs().max())), relu


def test_pack_weights_equals_convert_and_transpose_per_tensor():
    """mmamd_pack_weights: the bf16 copies and bf16 transposes of many fp32 matrices in one launch == mmamd_convert / mmamd_transpose_to_bf16 per
    tensor, bit for bit (ragged shapes, a zero tail in the transposes, more than 64 tensors = two launches)."""
    from multimodal_amd import ops

    g = torch.Generator().manual_seed(5)
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072), (96, 40), (130, 68), (64, 64), (1, 4)] + [(128, 256)] * 60
    ws = [torch.randn(r, c, generator=g).cuda() for r, c in shapes]
    nts, trs = ops.pack_weights(ws)
    for w, nt, tr in zip(ws, nts, trs):
        assert torch.equal(nt, ops.convert(w, torch.bfloat16))
        assert torch.equal(tr, ops.transpose_to_bf16(w, pad_to=64))
    only_t = ops.pack_weights(ws[:5], want_nt=False)
    assert only_t[0] is None and all(torch.equal(a, b) for a, b in zip(only_t[1], trs[:5]))


@pytest.mark.parametrize("B,S,H,causal,masked", [(3, 197, 12, False, False), (4, 77, 8, True, False), (2, 256, 4, False, True), (2, 50, 2, True, True),
                                                  (1, 1, 2, False, False), (2, 130, 3, False, True), (5, 33, 2, True, False)])
def test_fused_attention_backward_equals_the_two_kernel_form(B, S, H, causal, masked):
    """r04: ONE persistent kernel per launch (row images of a head staged once in LDS, whole-row loads and stores) replaces the dQ and dK/dV kernels
    for S <= 256 -- the two-role fused kernel (both roles per wave; causal shapes) and the single-pass kernel (every tile pair visited once, dS handed
    from the key tile's wave to the query tile's through an LDS mailbox; non-causal shapes of five tiles and more).  Same products and operand
    roundings as the two-kernel form (mmamd_debug_set_attn_variant(4000)); D = sum dO.O and the key-tile order of dQ are summed in another order, so the
    bf16 results agree to one unit in the last place, not bit for bit.  (Until r04's last day this test compared the fused kernel with itself: the A/B
    code 2000 belongs to the ring kernel's ablation range.)"""
    from multimodal_amd import _lib, ops

    torch.manual_seed(B * 1000 + S)
    qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
    dout = torch.randn(B * S, H * 64).to(torch.bfloat16).cuda()
    km = None
    if masked:
        km = (torch.rand(B, S) > 0.3).to(torch.uint8)
        km[:, 0] = 1
        km = km.cuda()
    out, lse = ops.attention_fwd_train(qkv, B, S, H, causal, km)
    L = _lib.lib()
    res = {}
    try:
        for code in (4000, 4001, 4002, 4003):  # two kernels | single pass (where built, else the fused one) | fused two-role | default
            L.mmamd_debug_set_attn_variant(code)
            res[code] = ops.attention_bwd(qkv, out, dout, lse, B, S, H, causal, km).float()
    finally:
        L.mmamd_debug_set_attn_variant(4003)
    ref = res[4000]
    for code in (4001, 4002, 4003):
        got = res[code]
        assert torch.isfinite(got).all()
        # one bf16 unit in the last place of the larger of the two values (2^-8 relative), plus the rounding of values near zero
        tol = 2.0 ** -7 * torch.maximum(got.abs(), ref.abs()) + 1e-3 * float(ref.abs().max())
        assert bool(((got - ref).abs() <= tol).all()), (code, float((got - ref).abs().max()))
    got = res[4003]
    again = ops.attention_bwd(qkv, out, dout, lse, B, S, H, causal, km).float()
    assert torch.equal(again, got)  # fixed summation order everywhere (mailbox schedule, no atomics): run-to-run bit-identical
    assert torch.isfinite(got.float()).all()


def test_grouped_weight_gradients_equal_the_single_launches_bit_for_bit():
    """r05: the four weight gradients of a layer in ONE split-K launch + one reduce launch (mmamd_gemm_bf16_tn_splitk_group): every problem's dW and db equal
    what mmamd_gemm_bf16_tn_splitk(_colsum) returns for it at the same number of splits, bit for bit (same kernel body, same partial order), on the shapes
    of a ViT-B/16 layer, a text layer and ragged ones (M, N not multiples of 256; a problem without a bias gradient; a single split)."""
    from multimodal_amd import ops

    set_rng_seed(23)
    cases = [
        (6400, [(768, 3072, True), (3072, 768, True), (768, 768, False), (2304, 768, True)], None),
        (2560, [(1536, 512, True), (512, 512, False), (2048, 512, True), (512, 2048, True)], None),
        (1280, [(200, 328, True), (72, 264, False), (520, 8, True)], 3),
        (256, [(256, 256, True), (264, 64, False)], 1),
        (1024, [(64, 64, False)] * 8, 4),
    ]
    for T, probs, splits in cases:
        jobs = [((torch.randn(T, M) * 0.1).to(torch.bfloat16).cuda(), torch.randn(T, N).to(torch.bfloat16).cuda(), cs) for M, N, cs in probs]
        tiles = sum(((M + 255) // 256) * ((N + 255) // 256) for M, N, _ in probs)
        s_used = splits if splits is not None else ops.wgrad_group_splits(tiles, T // 64)
        before = ops.launch_count("gemm_bf16_tn_splitk_group")
        outs = ops.gemm_bf16_tn_splitk_group(jobs, splits=splits)
        assert ops.launch_count("gemm_bf16_tn_splitk_group") == before + 1
        again = ops.gemm_bf16_tn_splitk_group(jobs, splits=splits)  # deterministic run to run
        for (y, x, cs), (dw, db), (dw2, db2) in zip(jobs, outs, again):
            assert torch.equal(dw, dw2) and (db is None or torch.equal(db, db2))
            if cs:
                ref_dw, ref_db = ops.gemm_bf16_tn_splitk(y, x, want_colsum=True, splits=s_used)
                assert torch.equal(db, ref_db), (T, tuple(y.shape), tuple(x.shape))
            else:
                ref_dw = ops.gemm_bf16_tn_splitk(y, x, splits=s_used)
                assert db is None
            assert torch.equal(dw, ref_dw), (T, tuple(y.shape), tuple(x.shape), s_used)
            want = y.double().T @ x.double()
            assert (dw.double() - want).abs().max().item() <= 2e-5 * T ** 0.5 * max(1.0, float(want.abs().max()))


def test_column_sums_of_few_very_wide_rows():
    """r05: mmamd_colsum's form for few rows of a very wide matrix (the positional-embedding gradient sums d_asm viewed as [B, S * w]): a thread per
    16-byte column chunk, eight row groups -- against float64 sums; shapes on both sides of its dispatch predicate, fp32 and bf16, a ragged row count."""
    from multimodal_amd import ops

    set_rng_seed(29)
    for rows, n, dt in ((256, 32768, torch.float32), (64, 16392, torch.bfloat16), (7, 20000, torch.float32), (203, 151296 // 8, torch.float32),
                        (300, 16384, torch.bfloat16), (256, 16376, torch.float32)):
        x = torch.randn(rows, n).to(dt)
        got = host(ops.colsum(x.cuda()))
        want = x.double().sum(0).numpy()
        assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), (rows, n, dt)


def test_f32_strided_gemm_with_the_contraction_split_over_the_waves():
    """r05: mmamd_f32_gemm_strided computes one 32 x 32 tile per workgroup, the contraction split over its four waves (partials added in wave order):
    against float64 at the shapes of the projection / loss gradients of a CLIP step, ragged shapes, short contractions (waves without any work) and both
    stride orders; deterministic run to run."""
    from multimodal_amd import ops

    set_rng_seed(31)
    for M, N, K in ((768, 512, 256), (256, 768, 512), (256, 512, 256), (40, 70, 5), (33, 31, 9), (64, 64, 1000), (1, 1, 1), (100, 3, 37)):
        X, Y = torch.randn(M, K), torch.randn(N, K)
        want = (X.double() @ Y.double().t()).numpy()
        C = ops.f32_gemm_strided(X.cuda(), K, 1, Y.cuda(), K, 1, M, N, K)
        assert np.abs(host(C) - want).max() <= 2e-6 * max(1.0, K ** 0.5) * max(1.0, np.abs(want).max()), (M, N, K)
        assert torch.equal(C, ops.f32_gemm_strided(X.cuda(), K, 1, Y.cuda(), K, 1, M, N, K))
        Xt, Yt = X.t().contiguous(), Y.t().contiguous()  # the same product from [K, M] / [K, N] buffers (element strides 1 and M / N)
        Ct = ops.f32_gemm_strided(Xt.cuda(), 1, M, Yt.cuda(), 1, N, M, N, K)
        assert torch.equal(Ct, C), (M, N, K)
