"""Training-time dropout / stochastic depth on the MI355X path (csrc/dropout.hip, _autograd.py).  The reference draws its masks from torch's
generator, so parity is stated per mask:
  (i)   p = 0 is the unchanged path;  (ii) every mask equals the numpy Philox restatement BIT FOR BIT, keep-rate and the 1 / (1 - p) scaling exact;
  (iii) with the masks known, forward and every gradient of an encoder stack equal the reference arithmetic applied with the same masks;
  (iv)  the reference's default fine-tuning model (flava_model_for_classification: classifier_dropout = 0.5) runs a train-mode step."""
import numpy as np
import pytest
import torch

from oracle import dropout_layers, philox

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,p,group", [(4096, 0.5, 0), (1 << 20, 0.1, 0), (12 * 4 * 64, 0.25, 4 * 64), (64, 0.0, 0), (1 << 16, 0.9, 0)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_kernel_mask_equals_oracle(n, p, group, dtype):
    from multimodal_amd import ops

    torch.manual_seed(0)
    x = torch.randn(n, device="cuda").to(dtype)
    seed, site = 0x1234_5678_9ABC_DEF, 37
    y, m = ops.dropout(x, p, seed, site, group=group, want_mask=True)
    want = philox.dropout_mask(n, p, seed, site, group)
    assert np.array_equal(m.cpu().numpy(), want)
    ref = philox.dropout_apply(x.float().cpu().numpy(), want, p)
    if dtype == torch.bfloat16:
        ref = torch.from_numpy(ref).to(torch.bfloat16).float().numpy()
    assert np.array_equal(y.float().cpu().numpy(), ref)
    if p > 0 and group == 0 and n >= 1 << 16:
        assert abs(float(m.float().mean()) - (1 - p)) < 5 * np.sqrt(p * (1 - p) / n)
    # residual form and the in-place form
    r = torch.randn(n, device="cuda")
    out = ops.dropout(x, p, seed, site, residual=r, group=group)
    assert np.array_equal(out.cpu().numpy(), (r.cpu().numpy() + philox.dropout_apply(x.float().cpu().numpy(), want, p)).astype(np.float32))
    x2 = x.clone()
    ops.dropout(x2, p, seed, site, group=group, out=x2)
    assert torch.equal(x2, y)


def test_dropout_fn_backward_uses_the_same_mask():
    from multimodal_amd._autograd import DropoutFn

    x = torch.randn(8, 96, device="cuda", requires_grad=True)
    y = DropoutFn.apply(x, 0.3, 99, 5, 0)
    g = torch.randn_like(y)
    y.backward(g)
    m = torch.from_numpy(philox.dropout_mask(x.numel(), 0.3, 99, 5).reshape(8, 96)).cuda().float()
    scale = float(np.float32(1) / (np.float32(1) - np.float32(0.3)))
    assert torch.equal(x.grad, torch.where(m != 0, g * scale, torch.zeros_like(g)))


def _encoder_pair(dropout, drop_path, n_layer=2, d=128, heads=2, ff=256):
    from multimodal_amd.modules.layers.transformer import TransformerEncoder

    torch.manual_seed(3)
    enc = TransformerEncoder(n_layer, d, heads, ff, dropout=dropout, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=True,
                             drop_path_rate=drop_path).cuda().train()
    layers = []
    for layer in enc.layer:
        lin1, lin2 = layer.feedforward.model[0], layer.feedforward.model[-1]
        f = lambda t: t.detach().cpu().clone().requires_grad_(True)  # noqa: E731
        layers.append({"Wqkv": f(layer.attention.input_proj.weight), "bqkv": f(layer.attention.input_proj.bias),
                       "Wo": f(layer.attention.output_proj.weight), "bo": f(layer.attention.output_proj.bias), "W1": f(lin1.weight),
                       "b1": f(lin1.bias), "W2": f(lin2.weight), "b2": f(lin2.bias), "g1": f(layer.attention_layernorm.weight),
                       "be1": f(layer.attention_layernorm.bias), "g2": f(layer.feedforward_layernorm.weight),
                       "be2": f(layer.feedforward_layernorm.bias), "eps1": layer.attention_layernorm.eps, "eps2": layer.feedforward_layernorm.eps})
    return enc, layers


@pytest.mark.parametrize("dropout,drop_path", [(0.1, None), (0.0, 0.4), (0.25, 0.3)])
def test_encoder_stack_with_known_masks(dropout, drop_path):
    enc, layers = _encoder_pair(dropout, drop_path)
    B, S, d = 8, 16, 128
    torch.manual_seed(11)
    x = torch.randn(B, S, d)
    gout = torch.randn(B, S, d)
    # the seed the forward will draw: same generator state, same draw (multimodal_amd/_autograd.py::draw_seed)
    torch.manual_seed(77)
    seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
    torch.manual_seed(77)
    xg = x.cuda().requires_grad_(True)
    y = enc(xg).last_hidden_state
    y.backward(gout.cuda())
    rates = None if drop_path is None else [float(v) for v in torch.linspace(0, drop_path, len(layers))]
    # a layer whose stochastic-depth rate is 0 keeps every sample (p = 0 -> identity); its branch dropout is replaced all the same
    xr = x.clone().requires_grad_(True)
    yr = dropout_layers.encoder_forward(xr, layers, 2, 0.0 if drop_path is not None else dropout, dropout, rates, seed)
    yr.backward(gout)
    scale = float(yr.detach().abs().max())
    assert float((y.detach().cpu() - yr.detach()).abs().max()) < 2e-2 * scale
    gscale = float(xr.grad.abs().max())
    assert float((xg.grad.cpu() - xr.grad).abs().max()) < 4e-2 * gscale
    for li, layer in enumerate(enc.layer):
        got = {"Wqkv": layer.attention.input_proj.weight.grad, "Wo": layer.attention.output_proj.weight.grad,
               "W1": layer.feedforward.model[0].weight.grad, "W2": layer.feedforward.model[-1].weight.grad,
               "b2": layer.feedforward.model[-1].bias.grad, "bo": layer.attention.output_proj.bias.grad, "g2": layer.feedforward_layernorm.weight.grad}
        for k, gten in got.items():
            ref = layers[li][k].grad
            assert float((gten.cpu() - ref).abs().max()) < 5e-2 * float(ref.abs().max()) + 1e-6, (li, k)
    # dropped samples / elements really are dropped: with p_branch > 0 the forward differs from the no-dropout forward
    enc.eval()
    with torch.no_grad():
        y0 = enc(x.cuda()).last_hidden_state
    assert float((y0 - y.detach()).abs().max()) > 1e-3


def test_zero_rates_take_the_unchanged_path():
    from multimodal_amd._autograd import stack_drop_spec

    enc, _ = _encoder_pair(0.0, None)
    assert stack_drop_spec(enc.layer) == ([], 0)
    enc2, _ = _encoder_pair(0.0, 0.0)  # stochastic depth with rate 0 everywhere
    assert stack_drop_spec(enc2.layer) == ([], 0)


def test_stochastic_depth_module_row_mode():
    from multimodal_amd.modules.layers.stochastic_depth import StochasticDepth

    sd = StochasticDepth(0.5, "row").cuda()
    x = torch.randn(64, 8, 32, device="cuda")
    sd.eval()
    assert sd(x) is x
    sd.train()
    y = sd(x)
    per_sample = (y.reshape(64, -1).abs().sum(1) == 0)
    assert 10 < int(per_sample.sum()) < 54                      # about half of the samples dropped ...
    kept = ~per_sample
    assert torch.equal(y[kept], x[kept] * 2.0)                  # ... survivors scaled by 1 / (1 - p), whole samples at a time
    with pytest.raises(ValueError):
        StochasticDepth(0.5, "column")


def test_flava_classification_default_model_trains():
    """reference models/flava/model.py:551: classifier_dropout = 0.5 is the DEFAULT of flava_model_for_classification."""
    from multimodal_amd.models.flava.model import flava_model_for_classification

    torch.manual_seed(0)
    small = dict(image_hidden_size=128, image_num_attention_heads=2, image_num_hidden_layers=2, image_intermediate_size=256, image_size=32,
                 patch_size=16, text_hidden_size=128, text_num_attention_heads=2, text_num_hidden_layers=2, text_intermediate_size=256,
                 vocab_size=200, max_position_embeddings=32, multimodal_hidden_size=128, multimodal_num_attention_heads=2,
                 multimodal_num_hidden_layers=2, multimodal_intermediate_size=256, text_and_image_proj_size=64)
    model = flava_model_for_classification(num_classes=3, classifier_in_dim=128, classifier_hidden_sizes=64, pretrained=False, **small).cuda().train()
    assert model.classifier.hidden_dropout_p() == 0.5
    image = torch.randn(4, 3, 32, 32, device="cuda")
    text = torch.randint(1, 200, (4, 16), device="cuda")
    labels = torch.tensor([0, 1, 2, 1], device="cuda")
    out = model(image=image, text=text, required_embedding="mm", labels=labels)
    out.loss.backward()
    grads = [p.grad for p in model.classifier.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    assert any(float(g.abs().max()) > 0 for g in grads)
    # two train-mode forwards differ (fresh masks), eval-mode forwards do not
    l1 = float(model(image=image, text=text, required_embedding="mm", labels=labels).loss)
    l2 = float(model(image=image, text=text, required_embedding="mm", labels=labels).loss)
    assert l1 != l2
    model.eval()
    with torch.no_grad():
        e1 = float(model(image=image, text=text, required_embedding="mm", labels=labels).loss)
        e2 = float(model(image=image, text=text, required_embedding="mm", labels=labels).loss)
    assert e1 == e2


def test_training_forward_hands_out_attached_hidden_states():
    """VERDICT r03 missing #4: the training forwards return every hidden state attached to the graph (reference transformer.py:230-247,
    flava/transformer.py:254-259).  A gradient that enters through hidden_states[1] of a 2-layer stack must equal the gradient of the same
    loss on a 1-layer stack with layer 0's weights; adding it to a loss on the result must give the sum of the two gradients."""
    import copy

    from multimodal_amd.modules.layers.transformer import TransformerEncoder

    torch.manual_seed(5)
    enc2 = TransformerEncoder(2, 128, 2, 256, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=True).cuda().train()
    enc1 = TransformerEncoder(1, 128, 2, 256, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=True).cuda().train()
    enc1.layer[0].load_state_dict(copy.deepcopy(enc2.layer[0].state_dict()))
    x = torch.randn(4, 16, 128, device="cuda")
    w = torch.randn(4, 16, 128, device="cuda")

    def grads(enc):
        return [p.grad.clone() for p in enc.layer[0].parameters()]

    out = enc2(x, return_hidden_states=True)
    hs = out.hidden_states
    assert len(hs) == 3 and hs[0] is x and all(h.requires_grad for h in hs[1:]) and hs[2] is out.last_hidden_state
    (hs[1] * w).sum().backward()
    g_mid = grads(enc2)
    assert all(p.grad is None or float(p.grad.abs().max()) == 0 for p in enc2.layer[1].parameters())  # layer 1 is not on that path
    o1 = enc1(x)
    (o1.last_hidden_state * w).sum().backward()
    for a, b in zip(g_mid, grads(enc1)):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-5 * float(b.abs().max() + 1))
    # both routes at once: d/dtheta [sum(h1 w) + sum(h2 w)] = the sum of the separate gradients
    enc2.zero_grad()
    out = enc2(x, return_hidden_states=True)
    (out.last_hidden_state * w).sum().backward()
    g_end = grads(enc2)
    enc2.zero_grad()
    out = enc2(x, return_hidden_states=True)
    ((out.hidden_states[1] * w).sum() + (out.last_hidden_state * w).sum()).backward()
    for got, a, b in zip(grads(enc2), g_mid, g_end):
        assert float((got - (a + b)).abs().max()) <= 2e-2 * float((a + b).abs().max()) + 1e-6


def test_flava_training_forward_returns_attentions_and_hidden_states():
    from multimodal_amd.models.flava.model import flava_model
    from multimodal_amd.schedule import get_schedule, set_schedule

    torch.manual_seed(0)
    small = dict(image_hidden_size=128, image_num_attention_heads=2, image_num_hidden_layers=2, image_intermediate_size=256, image_size=32,
                 patch_size=16, text_hidden_size=128, text_num_attention_heads=2, text_num_hidden_layers=2, text_intermediate_size=256,
                 vocab_size=200, max_position_embeddings=32, multimodal_hidden_size=128, multimodal_num_attention_heads=2,
                 multimodal_num_hidden_layers=2, multimodal_intermediate_size=256, text_and_image_proj_size=64)
    model = flava_model(**small).cuda()
    image = torch.randn(4, 3, 32, 32, device="cuda")
    text = torch.randint(1, 200, (4, 16), device="cuda")
    model.eval()
    with torch.no_grad():
        ref = model(image=image, text=text, skip_unmasked_mm_encoder=False)
    model.train()
    out = model(image=image, text=text, skip_unmasked_mm_encoder=False)
    for name in ("image", "text", "multimodal"):
        tr, ev = getattr(out, name), getattr(ref, name)
        assert len(tr.hidden_states) == len(ev.hidden_states) == 3 and len(tr.attentions) == len(ev.attentions) == 2
        assert all(h.requires_grad for h in tr.hidden_states[1:])
        for a, b in zip(tr.attentions, ev.attentions):
            assert a.shape == b.shape and a.dtype == b.dtype and not a.requires_grad
            assert float((a - b).abs().max()) < 2e-3  # same weights, same inputs: the eval-mode probabilities
        for a, b in zip(tr.hidden_states, ev.hidden_states):
            assert float((a.detach() - b).abs().max()) < 2e-2 * float(b.abs().max())
    prev = get_schedule()
    try:
        set_schedule(train_attentions=False)
        assert model(image=image, text=text).image.attentions is None
    finally:
        set_schedule(train_attentions=prev.train_attentions)


@pytest.mark.parametrize("hm_kind", ["none", "per_head", "full"])
def test_attention_probability_dropout_kernels_vs_oracle(hm_kind):
    """mmamd_attention_x_fwd_dropout / _bwd_dropout: the returned (dropped) probabilities equal softmax * Philox mask / (1 - p) element for
    element, O = P' V, and dQ / dK / dV equal torch autograd of the same expression with the same mask.  With a head_mask as well
    (mmamd_attention_x_fwd/_bwd_dropout_head_mask, r06): P' = P keep / (1 - p) m, the order of the reference (modules/layers/attention.py:232-237:
    F.dropout, then `attn = attn * head_mask`) -- a per-head [1, H, 1, 1] mask and a full [B, H, S, S] one."""
    from multimodal_amd import ops

    B, H, S, hd, p, seed, site = 2, 2, 40, 64, 0.2, 4242, 19
    torch.manual_seed(1)
    q, k, v = (torch.randn(B * S, H * hd).to(torch.bfloat16) for _ in range(3))
    km = torch.ones(B, S, dtype=torch.uint8)
    km[1, 33:] = 0
    hmask = {"none": None, "per_head": torch.tensor([0.5, 1.5]).view(1, H, 1, 1), "full": torch.rand(B, H, S, S) * 2}[hm_kind]
    hm_dev = hmask.cuda() if hmask is not None else None
    lse = torch.empty((B, H, S), dtype=torch.float32, device="cuda")
    out, probs = ops.attention_x_fwd(q.cuda(), k.cuda(), v.cuda(), B, S, S, H, hd, ops.AttnMask(key_mask=km.cuda()), want_probs=True, lse=lse,
                                     drop=(p, seed, site), head_mask=hm_dev)
    keep = torch.from_numpy(dropout_layers.attention_mask_bhqk(B, H, S, S, p, seed, site))
    scale = float(np.float32(1) / (np.float32(1) - np.float32(p)))

    def ref(qf, kf, vf):
        qh, kh, vh = (t.view(B, S, H, hd).transpose(1, 2) for t in (qf, kf, vf))
        s = (qh @ kh.transpose(-1, -2)) / 8.0
        s = s.masked_fill(km[:, None, None, :] == 0, float("-inf"))
        pr = torch.softmax(s, -1) * keep * scale
        if hmask is not None:
            pr = pr * hmask
        return pr, (pr @ vh).transpose(1, 2).reshape(B * S, H * hd)

    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    pr, o = ref(qf, kf, vf)
    assert float((probs.cpu() - pr.detach()).abs().max()) < 2e-3
    assert torch.equal(probs.cpu() == 0, pr.detach() == 0) or float(((probs.cpu() == 0) != (pr.detach() == 0)).float().mean()) < 1e-4  # same dropped set
    assert float((out.float().cpu() - o.detach()).abs().max()) < 3e-2
    do = torch.randn(B * S, H * hd).to(torch.bfloat16)
    o.backward(do.float())
    dq, dkv = ops.attention_x_bwd(q.cuda(), k.cuda(), v.cuda(), out, do.cuda(), lse, B, S, S, H, hd, ops.AttnMask(key_mask=km.cuda()),
                                  drop=(p, seed, site), head_mask=hm_dev)
    D = H * hd
    for got, want, name in ((dq, qf.grad, "dq"), (dkv[:, :D], kf.grad, "dk"), (dkv[:, D:], vf.grad, "dv")):
        err = float((got.float().cpu() - want).abs().max())
        assert err < 4e-2 * float(want.abs().max()) + 1e-3, (name, err)


def test_flava_encoder_with_dropout_on_every_site():
    """FLAVA's encoder built with dropout > 0 (models/flava/transformer.py:87-90 puts the same rate on the attention probabilities, both residual
    branches and the MLP): forward and gradients vs the reference arithmetic with the Philox masks of the four sites."""
    from multimodal_amd.models.flava.transformer import TransformerEncoder

    p, d, H = 0.15, 128, 2
    torch.manual_seed(4)
    enc = TransformerEncoder(2, d, H, 256, dropout=p, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=True).cuda().train()
    layers = []
    for layer in enc.layer:
        at = layer.attention
        f = lambda t: t.detach().cpu().clone().requires_grad_(True)  # noqa: E731
        layers.append({"Wqkv": f(torch.cat([at.query.weight, at.key.weight, at.value.weight])), "bqkv": f(torch.cat([at.query.bias, at.key.bias, at.value.bias])),
                       "Wo": f(at.output.weight), "bo": f(at.output.bias), "W1": f(layer.feedforward.model[0].weight), "b1": f(layer.feedforward.model[0].bias),
                       "W2": f(layer.feedforward.model[-1].weight), "b2": f(layer.feedforward.model[-1].bias), "g1": f(layer.attention_layernorm.weight),
                       "be1": f(layer.attention_layernorm.bias), "g2": f(layer.feedforward_layernorm.weight), "be2": f(layer.feedforward_layernorm.bias),
                       "eps1": layer.attention_layernorm.eps, "eps2": layer.feedforward_layernorm.eps})
    B, S = 4, 24
    x, gout = torch.randn(B, S, d), torch.randn(B, S, d)
    torch.manual_seed(123)
    seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
    torch.manual_seed(123)
    xg = x.cuda().requires_grad_(True)
    out = enc(xg, return_hidden_states=True)
    out.last_hidden_state.backward(gout.cuda())
    xr = x.clone().requires_grad_(True)
    yr = dropout_layers.flava_encoder_forward(xr, layers, H, p, seed)
    yr.backward(gout)
    assert float((out.last_hidden_state.detach().cpu() - yr.detach()).abs().max()) < 2e-2 * float(yr.detach().abs().max())
    assert float((xg.grad.cpu() - xr.grad).abs().max()) < 5e-2 * float(xr.grad.abs().max())
    for li, layer in enumerate(enc.layer):
        d_ = d
        gq = torch.cat([layer.attention.query.weight.grad, layer.attention.key.weight.grad, layer.attention.value.weight.grad]).cpu()
        for got, want, name in ((gq, layers[li]["Wqkv"].grad, "Wqkv"), (layer.attention.output.weight.grad.cpu(), layers[li]["Wo"].grad, "Wo"),
                                (layer.feedforward.model[0].weight.grad.cpu(), layers[li]["W1"].grad, "W1")):
            assert float((got - want).abs().max()) < 6e-2 * float(want.abs().max()) + 1e-6, (li, name)


def test_flava_encoder_head_mask_with_dropout_trains():
    """head_mask TOGETHER with attention dropout in training (raised until r06; the reference applies F.dropout and then `attn * head_mask`,
    modules/layers/attention.py:232-237).  Module level: (a) an all-ones head_mask is the identity -- the step equals the one without a mask under the
    same seed, output and every gradient bit for bit (x 1.0 is exact; the kernels regenerate the same Philox masks); (b) zeroing head 0 of every layer
    equals zeroing the corresponding 64 input columns of each layer's output projection."""
    from multimodal_amd.models.flava.transformer import TransformerEncoder

    p, d, H, B, S = 0.2, 128, 2, 3, 40
    torch.manual_seed(7)
    enc = TransformerEncoder(2, d, H, 256, dropout=p, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=True).cuda().train()
    x, gout = torch.randn(B, S, d).cuda(), torch.randn(B, S, d).cuda()

    def step(head_mask):
        enc.zero_grad(set_to_none=True)
        torch.manual_seed(99)  # the stack draws its Philox seed from torch's generator
        xg = x.clone().requires_grad_(True)
        out = enc(xg, head_mask=head_mask, return_hidden_states=True).last_hidden_state
        out.backward(gout)
        return out.detach().clone(), xg.grad.clone(), {n: q.grad.clone() for n, q in enc.named_parameters()}

    o0, gx0, gp0 = step(None)
    o1, gx1, gp1 = step(torch.ones(1, H, 1, 1, device="cuda"))
    assert torch.equal(o0, o1) and torch.equal(gx0, gx1)
    assert all(torch.equal(gp0[n], gp1[n]) for n in gp0)
    hm = torch.ones(1, H, 1, 1, device="cuda")
    hm[0, 0] = 0.0
    o2, gx2, _ = step(hm)
    saved = [layer.attention.output.weight.detach().clone() for layer in enc.layer]
    with torch.no_grad():
        for layer in enc.layer:
            layer.attention.output.weight[:, :64] = 0.0  # head 0's slice of the merged heads
    o3, gx3, _ = step(None)
    with torch.no_grad():
        for layer, w in zip(enc.layer, saved):
            layer.attention.output.weight.copy_(w)
    assert float((o2 - o3).abs().max()) <= 2e-2 * float(o3.abs().max())
    assert float((gx2 - gx3).abs().max()) <= 5e-2 * float(gx3.abs().max())
    assert float((o2 - o0).abs().max()) > 1e-2 * float(o0.abs().max())  # (the mask did something)


def test_decoder_stack_trains_with_dropout():
    """TransformerDecoder (CoCa's text / multimodal decoders) with dropout > 0: all six sites per layer (self- and cross-attention
    probabilities, three residual branches, the MLP) run; eval == dropout-free; two training forwards differ; p -> 0 converges to the
    dropout-free gradients (the masks keep everything when p is tiny: same arithmetic through the dropout code path)."""
    import copy

    from multimodal_amd.modules.layers.transformer import TransformerDecoder

    torch.manual_seed(8)
    kw = dict(n_layer=2, d_model=128, n_head=2, dim_feedforward=256, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=True,
              use_cross_attention=True, dim_kv=128)
    dec0 = TransformerDecoder(dropout=0.0, **kw).cuda().train()
    decp = TransformerDecoder(dropout=1e-9, **kw).cuda().train()   # takes the dropout code path, drops (almost surely) nothing
    decd = TransformerDecoder(dropout=0.3, **kw).cuda().train()
    sd = copy.deepcopy(dec0.state_dict())
    for m in (decp, decd):
        # Dropout modules shift the MLP's Sequential indices: feedforward.model.2 -> .3
        m.load_state_dict({k.replace("feedforward.model.2.", "feedforward.model.3."): v for k, v in sd.items()}, strict=True)
    x = torch.randn(3, 12, 128, device="cuda")
    enc = torch.randn(3, 9, 128, device="cuda")
    causal = torch.ones(12, 12, dtype=torch.bool, device="cuda").tril()
    w = torch.randn(3, 12, 128, device="cuda")

    def step(m):
        m.zero_grad()
        y = m(x, enc, attention_mask=causal).last_hidden_state
        (y * w).sum().backward()
        return y.detach(), [p.grad.clone() for p in m.parameters()]

    y0, g0 = step(dec0)
    yp, gp = step(decp)
    assert float((y0 - yp).abs().max()) < 1e-3 * float(y0.abs().max()) + 1e-4
    for a, b in zip(g0, gp):
        assert float((a - b).abs().max()) <= 2e-2 * float(a.abs().max()) + 1e-5
    y1, g1 = step(decd)
    y2, _ = step(decd)
    assert float((y1 - y2).abs().max()) > 1e-3 and float((y1 - y0).abs().max()) > 1e-3   # fresh masks per forward
    assert all(torch.isfinite(g).all() for g in g1)
    decd.eval()
    with torch.no_grad():
        ye = decd(x, enc, attention_mask=causal).last_hidden_state
    assert float((ye - y0).abs().max()) < 2e-2 * float(y0.abs().max())   # eval: dropout is the identity (inference kernels vs training forward)


def test_decoder_training_forward_returns_attached_hidden_states():
    """TransformerDecoder(..., return_hidden_states=True) in training (reference transformer.py:606-640 returns the input, every layer's output and keeps
    them in the graph): the per-layer-node form gives the same last state and the same parameter gradients as the one-node form, a loss on an
    INTERMEDIATE state reaches only the layers below it, and with dropout both forms draw the same masks from the same seed (layer0 keeps the sites)."""
    from multimodal_amd.modules.layers.transformer import TransformerDecoder

    torch.manual_seed(13)
    kw = dict(n_layer=3, d_model=128, n_head=2, dim_feedforward=256, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=True,
              use_cross_attention=True, dim_kv=128)
    dec = TransformerDecoder(dropout=0.0, **kw).cuda().train()
    x = torch.randn(2, 10, 128, device="cuda")
    enc = torch.randn(2, 7, 128, device="cuda")
    causal = torch.ones(10, 10, dtype=torch.bool, device="cuda").tril()
    w = torch.randn(2, 10, 128, device="cuda")

    def grads():
        return [p.grad.clone() if p.grad is not None else None for p in dec.parameters()]

    dec.zero_grad()
    one = dec(x, enc, attention_mask=causal)
    (one.last_hidden_state * w).sum().backward()
    g_one = grads()
    assert one.hidden_states == []
    dec.zero_grad()
    per = dec(x, enc, attention_mask=causal, return_hidden_states=True)
    assert len(per.hidden_states) == 4 and all(h.shape == (2, 10, 128) for h in per.hidden_states)
    assert torch.equal(per.hidden_states[0], x) and torch.equal(per.hidden_states[-1], per.last_hidden_state)
    assert torch.equal(per.last_hidden_state, one.last_hidden_state)   # the same kernels on the same data
    assert all(h.requires_grad for h in per.hidden_states[1:])
    (per.last_hidden_state * w).sum().backward()
    for a, b in zip(g_one, grads()):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b))
    # a loss on the state after layer 0 reaches layer 0's parameters only
    dec.zero_grad()
    per = dec(x, enc, attention_mask=causal, return_hidden_states=True)
    (per.hidden_states[1] * w).sum().backward()
    got = {n: p.grad is not None and bool((p.grad != 0).any()) for n, p in dec.named_parameters()}
    assert all(v for n, v in got.items() if n.startswith("layer.0.")) and not any(v for n, v in got.items() if not n.startswith("layer.0."))
    # dropout: same seed -> same masks in both forms
    decd = TransformerDecoder(dropout=0.25, **kw).cuda().train()
    torch.manual_seed(99)
    a = decd(x, enc, attention_mask=causal).last_hidden_state.detach()
    torch.manual_seed(99)
    b = decd(x, enc, attention_mask=causal, return_hidden_states=True).last_hidden_state.detach()
    assert torch.equal(a, b)


def test_eval_mode_stack_with_grad_input_applies_no_dropout():
    """a frozen .eval() stack fed by a trainable upstream module lands on the differentiable path because its INPUT
    requires grad; nn.Dropout / StochasticDepth are the identity in eval mode (reference transformer.py:64-93 builds plain nn.Dropout /
    StochasticDepth modules), so the result must equal the no-grad inference forward, twice in a row, and the input gradient must be
    that of the dropout-free stack."""
    from multimodal_amd._autograd import stack_drop_spec
    from multimodal_amd.models.flava.transformer import TransformerEncoder as FlavaEncoder
    from multimodal_amd.modules.layers.transformer import TransformerDecoder

    enc, _ = _encoder_pair(0.4, 0.3)
    assert stack_drop_spec(enc.layer)[0] != [] and stack_drop_spec(enc.layer, training=False) == ([], 0)
    torch.manual_seed(11)
    dec = TransformerDecoder(n_layer=2, d_model=128, n_head=2, dim_feedforward=256, dropout=0.3, activation=torch.nn.GELU,
                             layer_norm_eps=1e-5, norm_first=True, use_cross_attention=False).cuda()
    fl = FlavaEncoder(n_layer=2, d_model=128, n_head=2, dim_feedforward=256, dropout=0.3, activation=torch.nn.GELU, layer_norm_eps=1e-5,
                      norm_first=True).cuda()   # SelfAttention(dropout): attention-probability dropout too
    for name, mod in (("encoder", enc), ("decoder", dec), ("flava", fl)):
        mod.eval()
        for p in mod.parameters():
            p.requires_grad_(False)
        x = torch.randn(3, 12, 128, device="cuda")
        with torch.no_grad():
            y0 = mod(x).last_hidden_state
        xg = x.clone().requires_grad_(True)
        y1 = mod(xg).last_hidden_state
        y2 = mod(xg).last_hidden_state
        assert y1.requires_grad, name
        assert torch.equal(y1.detach(), y2.detach()), name                      # no fresh masks between two forwards
        assert float((y1.detach() - y0).abs().max()) < 2e-2 * float(y0.abs().max()), name   # training forward vs inference kernels, bf16
        (y1 * y1.detach()).sum().backward()
        assert torch.isfinite(xg.grad).all() and float(xg.grad.abs().max()) > 0, name
