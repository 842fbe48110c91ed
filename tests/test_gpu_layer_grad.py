"""Training-path holes of r04 closed in r05, each pinned to gradients the REFERENCE's torch autograd produced
(tests/golden/make_golden_layer_grad.py -> layer_grad.npz; VERDICT r04 missing 4 / next 6):
  * post-norm layers -- the reference's DEFAULT, norm_first=False (modules/layers/transformer.py:56,118-132; flava/transformer.py:178-198);
  * boolean attention masks in a TRAINING TransformerEncoder (causal [S, S] and arbitrary [B, 1, S, S]);
  * a stand-alone TransformerEncoderLayer called in training (what an FSDP / checkpoint-wrapped layer runs);
  * a stack whose layers carry hooks is served layer by layer through the layers' own forwards, with the same results.
Tolerances: those of the stack-level gradient tests (bf16 MFMA operands, fp32 accumulation)."""
import numpy as np
import pytest
import torch

from tests._util import fixture_sd

pytestmark = pytest.mark.gpu


def host(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def _check(z, tag, mod, y, xg, y_tol=2e-2, sd_tag=None):
    ref_y = z[f"{tag}.y"].astype(np.float64)
    assert np.abs(host(y) - ref_y).max() <= y_tol * max(1.0, np.abs(ref_y).max()), (tag, np.abs(host(y) - ref_y).max())
    ref_dx = z[f"{tag}.dx"].astype(np.float64)
    assert np.abs(host(xg.grad) - ref_dx).max() <= 6e-2 * np.abs(ref_dx).max(), (tag, "dx")
    worst = ("", 0.0)
    gscale = max(float(np.abs(z[f"{tag}.g.{k}"]).max()) for k, _ in mod.named_parameters())  # largest gradient entry of the module
    for k, p in mod.named_parameters():
        ref = z[f"{tag}.g.{k}"].astype(np.float64)
        assert p.grad is not None, (tag, k)
        got = host(p.grad)
        if np.abs(ref).max() < 1e-6 * gscale:
            # mathematically zero (key biases: a shift of every score of a query is softmax-invariant): the reference holds fp32 round-off
            # there, this path the bf16 round-off of the dK rows it sums -- both ~0 on the scale of the module's gradients
            assert k.endswith("key.bias") or k.endswith("k_proj.bias") or "in_proj" in k or k.endswith("input_proj.bias"), (tag, k)
            assert np.abs(got).max() <= 1e-3 * gscale, (tag, k, np.abs(got).max(), gscale)
            continue
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        rms = np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-12)
        assert rel <= 6e-2 and rms <= 3e-2, (tag, k, rel, rms)
        if rel > worst[1]:
            worst = (k, rel)
    print(tag, "worst max-rel", worst)


def _load(mod, z, tag):
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in fixture_sd(z, prefix=f"{tag}.sd.").items()}, strict=True)
    return mod.cuda().train()


def _step(mod, z, tag, call):
    x = torch.from_numpy(z[f"{tag}.x"]).cuda().requires_grad_(True)
    w = torch.from_numpy(z[f"{tag}.w"]).cuda()
    mod.zero_grad()
    y = call(mod, x)
    (y * w).sum().backward()
    return y, x


def test_post_norm_encoder_trains_like_the_reference(golden):
    from multimodal_amd.modules.layers.transformer import TransformerEncoder

    z = golden("layer_grad.npz")
    enc = _load(TransformerEncoder(n_layer=2, d_model=128, n_head=2, dim_feedforward=128, activation=torch.nn.GELU, layer_norm_eps=1e-5,
                                   norm_first=False, final_layer_norm_eps=1e-5), z, "post")
    y, x = _step(enc, z, "post", lambda m, t: m(t).last_hidden_state)
    _check(z, "post", enc, y, x)
    # hidden states of the training forward: the input, each layer's (normalised) output -- attached to the graph
    out = enc(torch.from_numpy(z["post.x"]).cuda(), return_hidden_states=True)
    assert len(out.hidden_states) == 3 and out.hidden_states[1].requires_grad
    for got, ref in zip(out.hidden_states, z["post.hidden"]):
        assert np.abs(host(got) - ref).max() <= 2e-2 * max(1.0, np.abs(ref).max())
    # eval mode (inference kernels) agrees with the training forward
    enc.eval()
    with torch.no_grad():
        ye = enc(torch.from_numpy(z["post.x"]).cuda()).last_hidden_state
    assert float((ye - y.detach()).abs().max()) <= 2e-2 * float(y.detach().abs().max())


def test_flava_post_norm_encoder_with_key_padding_trains_like_the_reference(golden):
    from multimodal_amd.models.flava.transformer import TransformerEncoder

    z = golden("layer_grad.npz")
    enc = _load(TransformerEncoder(n_layer=1, d_model=128, n_head=2, dim_feedforward=128, activation=torch.nn.GELU, layer_norm_eps=1e-5,
                                   norm_first=False), z, "fpost")
    am = torch.from_numpy(z["fpost.mask"]).cuda()
    y, x = _step(enc, z, "fpost", lambda m, t: m(t, attention_mask=am).last_hidden_state)
    _check(z, "fpost", enc, y, x)


def test_training_encoder_under_boolean_attention_masks(golden):
    from multimodal_amd.modules.layers.transformer import TransformerEncoder

    z = golden("layer_grad.npz")
    enc = _load(TransformerEncoder(n_layer=2, d_model=128, n_head=2, dim_feedforward=128, activation=torch.nn.GELU, layer_norm_eps=1e-5,
                                   norm_first=True), z, "mask.causal")
    causal = torch.ones(10, 10, dtype=torch.bool, device="cuda").tril()
    y, x = _step(enc, z, "mask.causal", lambda m, t: m(t, attention_mask=causal).last_hidden_state)
    _check(z, "mask.causal", enc, y, x)
    rnd = torch.from_numpy(z["mask.rnd.mask"]).cuda().unsqueeze(1)
    y, x = _step(enc, z, "mask.rnd", lambda m, t: m(t, attention_mask=rnd).last_hidden_state)
    _check(z, "mask.rnd", enc, y, x)


def test_stand_alone_layers_are_differentiable(golden):
    from multimodal_amd.models.flava.transformer import TransformerEncoderLayer as FlavaLayer
    from multimodal_amd.modules.layers.transformer import TransformerEncoderLayer

    z = golden("layer_grad.npz")
    lay = _load(TransformerEncoderLayer(d_model=128, n_head=2, dim_feedforward=128, activation=torch.nn.GELU, layer_norm_eps=1e-5,
                                        norm_first=True), z, "lone.pre")
    y, x = _step(lay, z, "lone.pre", lambda m, t: m(t))
    assert y.requires_grad
    _check(z, "lone.pre", lay, y, x)
    flay = _load(FlavaLayer(d_model=128, n_head=2, dim_feedforward=128, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=False), z,
                 "lone.fpost")
    y, x = _step(flay, z, "lone.fpost", lambda m, t: m(t))
    _check(z, "lone.fpost", flay, y, x)
    # with return_attn_weights the training forward hands out (y, probabilities) like the reference
    y2, probs = flay(torch.from_numpy(z["lone.fpost.x"]).cuda(), return_attn_weights=True)
    assert probs.shape == (2, 2, 9, 9) and float((probs.sum(-1) - 1).abs().max()) < 1e-3
    assert float((y2.detach() - y.detach()).abs().max()) == 0.0


@pytest.mark.parametrize("family", ["generic", "flava", "decoder"])
def test_hooked_layers_are_called_one_by_one_with_the_same_results(family):
    """A stack whose layers carry hooks (or wrappers: FSDP, checkpoint_wrapper) must CALL its layers (plain_layers() is False): same
    forward values and gradients as the stack-level node, and the hooks fire once per layer and forward."""
    import copy

    from multimodal_amd._autograd import plain_layers

    torch.manual_seed(5)
    if family == "generic":
        from multimodal_amd.modules.layers.transformer import TransformerEncoder as Enc, TransformerEncoderLayer as Lay

        enc = Enc(n_layer=3, d_model=128, n_head=2, dim_feedforward=256, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=True,
                  final_layer_norm_eps=1e-5).cuda().train()
        call = lambda m, t: m(t, return_hidden_states=True)
    elif family == "flava":
        from multimodal_amd.models.flava.transformer import TransformerEncoder as Enc, TransformerEncoderLayer as Lay

        enc = Enc(n_layer=3, d_model=128, n_head=2, dim_feedforward=256, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=True).cuda().train()
        call = lambda m, t: m(t, return_hidden_states=True, return_attn_weights=True)
    else:
        from multimodal_amd.modules.layers.transformer import TransformerDecoder as Enc, TransformerDecoderLayer as Lay

        enc = Enc(n_layer=2, d_model=128, n_head=2, dim_feedforward=256, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=True,
                  use_cross_attention=True, dim_kv=128).cuda().train()
        mem = torch.randn(3, 9, 128, device="cuda")
        cm = torch.ones(12, 12, dtype=torch.bool, device="cuda").tril()
        call = lambda m, t: m(t, mem, attention_mask=cm, return_hidden_states=True)
    hooked = copy.deepcopy(enc)
    fired = []
    for i, layer in enumerate(hooked.layer):
        layer.register_forward_hook(lambda mod, args, out, i=i: fired.append(i))
    assert plain_layers(enc.layer, Lay) and not plain_layers(hooked.layer, Lay)
    x = torch.randn(3, 12, 128, device="cuda")
    w = torch.randn(3, 12, 128, device="cuda")

    def step(m):
        xg = x.clone().requires_grad_(True)
        o = call(m, xg)
        (o.last_hidden_state * w).sum().backward()
        return o, xg.grad, [p.grad.clone() for p in m.parameters()]

    o1, dx1, g1 = step(enc)
    o2, dx2, g2 = step(hooked)
    assert fired == list(range(len(hooked.layer)))
    assert float((o1.last_hidden_state - o2.last_hidden_state).abs().max()) <= 1e-5 * float(o1.last_hidden_state.abs().max())
    assert len(o1.hidden_states) == len(o2.hidden_states)
    for a, b in zip(o1.hidden_states, o2.hidden_states):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    # gradients: the same kernels; the one difference is where a bias gradient's column sums come from (the stack-level node takes two of them
    # from the fp32 dX inside the LayerNorm backward of the layer above, a stand-alone layer from the bf16 dX it was handed): bf16 rounding level
    assert float((dx1 - dx2).abs().max()) <= 2e-2 * float(dx1.abs().max()), float((dx1 - dx2).abs().max()) / float(dx1.abs().max())
    for (name, _), a, b in zip(enc.named_parameters(), g1, g2):
        assert float((a - b).abs().max()) <= 2e-2 * float(a.abs().max()) + 1e-6, (name, float((a - b).abs().max()), float(a.abs().max()))
    # inference through hooked layers too
    fired.clear()
    enc.eval(); hooked.eval()
    with torch.no_grad():
        e1, e2 = call(enc, x), call(hooked, x)
    assert fired == list(range(len(hooked.layer)))
    assert float((e1.last_hidden_state - e2.last_hidden_state).abs().max()) <= 1e-5 * float(e1.last_hidden_state.abs().max())


def test_head_mask_in_training_matches_the_reference_gradients(golden):
    """r05: the reference's `head_mask` while TRAINING (modules/layers/attention.py:236-237: multiplied into the probabilities
    after softmax and dropout; flava/transformer.py:268-275: the same mask for every layer).  The general attention kernels carry it in the forward
    and in both backward kernels (P' = P m in dV, dP m in dS).  Fixture: tests/golden/make_golden_head_mask.py -> head_mask_grad.npz -- a 2-layer
    pre-norm FLAVA encoder in train mode, real-valued [2, 2, 9, 9] mask with one head of one sample pruned, key-padding mask; the input gradient and
    all 32 parameter gradients from the reference's torch autograd; also through stand-alone layers (one autograd node per layer)."""
    from multimodal_amd.models.flava.transformer import TransformerEncoder

    z = golden("head_mask_grad.npz")
    t = lambda k: torch.from_numpy(z[k]).cuda()  # noqa: E731
    for per_layer in (False, True):
        enc = TransformerEncoder(n_layer=2, d_model=128, n_head=2, dim_feedforward=256, activation=torch.nn.GELU, norm_first=True)
        enc.load_state_dict({k[len("sd."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}, strict=True)
        enc = enc.cuda().train()
        x = t("x").requires_grad_(True)
        if per_layer:  # a user's own loop over the layers: every layer is its own autograd node
            h = x
            for layer in enc.layer:
                h = layer(h, attention_mask=t("mask"), head_mask=t("hm"))
            last, attn = h, None
        else:
            o = enc(x, attention_mask=t("mask"), head_mask=t("hm"), return_attn_weights=True, return_hidden_states=True)
            last, attn = o.last_hidden_state, o.attentions
        loss = (last * t("w")).sum()
        loss.backward()
        assert abs(float(loss) - float(z["loss"])) <= 2e-2 * max(1.0, abs(float(z["loss"])))
        assert np.abs(host(last) - z["last"]).max() <= 3e-2
        if attn is not None:  # the returned maps carry the mask (values, detached)
            for i in range(2):
                assert np.abs(host(attn[i]) - z["attn"][i]).max() <= 5e-3, i
            assert float(host(attn[0])[0, 1].max()) == 0.0  # the pruned head of sample 0
        ref_dx = z["dx"].astype(np.float64)
        assert np.abs(host(x.grad) - ref_dx).max() <= 6e-2 * np.abs(ref_dx).max()
        gscale = max(float(np.abs(z["g." + k]).max()) for k, _ in enc.named_parameters())
        worst = ("", 0.0)
        for k, p in enc.named_parameters():
            ref = z["g." + k].astype(np.float64)
            got = host(p.grad)
            if np.abs(ref).max() < 1e-6 * gscale:  # key biases: mathematically zero (see _check)
                assert k.endswith("key.bias") and np.abs(got).max() <= 1e-3 * gscale, (k, np.abs(got).max())
                continue
            rel = np.abs(got - ref).max() / np.abs(ref).max()
            rms = np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-12)
            assert rel <= 6e-2 and rms <= 3e-2, (per_layer, k, rel, rms)
            worst = (k, rel) if rel > worst[1] else worst
        print("head_mask training, per_layer =", per_layer, "worst max-rel", worst)


def test_interpolate_pos_encoding_in_training_matches_the_reference_gradients(golden):
    """r05: ImageEmbeddings(..., interpolate_pos_encoding=True) while TRAINING (reference models/flava/image_encoder.py:102-137,170-173) -- the bicubic
    resampling of the position table is linear in the table, its backward the transpose of the same map (models/flava/_train.py::BicubicTableFn).
    Fixture: tests/golden/make_golden_interp_grad.py -> interp_grad.npz (4 x 4 grid resampled to 6 x 6, patch mask, all five parameter gradients from the
    reference's torch autograd through F.interpolate(mode="bicubic"))."""
    from multimodal_amd.models.flava.image_encoder import ImageEmbeddings

    z = golden("interp_grad.npz")
    emb = ImageEmbeddings(image_size=64, patch_size=16, hidden_size=128, use_image_masking=True)
    emb.load_state_dict({k[len("sd."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}, strict=True)
    emb = emb.cuda().train()
    t = lambda k: torch.from_numpy(z[k]).cuda()  # noqa: E731
    out = emb(t("image"), image_patches_mask=t("patches_mask"), interpolate_pos_encoding=True)
    assert out.shape == (3, 37, 128) and out.grad_fn is not None
    assert np.abs(host(out) - z["out"]).max() <= 2e-2 * max(1.0, np.abs(z["out"]).max())  # patch GEMM in bf16
    (out * t("w")).sum().backward()
    for k, p in emb.named_parameters():
        ref = z["g." + k].astype(np.float64)
        got = host(p.grad)
        assert got.shape == ref.shape, k
        rel = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)
        # position / cls gradients are exact-fp32 sums of w (no bf16 operand on their path: the resampling's transpose runs on the exact-fp32 MFMA);
        # the projection's come from bf16-operand GEMMs, and the mask token's is (all patch rows) - (the conv-bias gradient), which carries that rounding
        assert rel <= (6e-2 if "projection" in k else 1e-2 if k == "mask_token" else 1e-5), (k, rel)
    with pytest.raises(ValueError):
        emb(t("image"))  # without the flag the reference's size check stands, in training too


def test_clip_text_encoder_hidden_states_in_training_match_the_reference_gradients(golden):
    """r05: CLIPTextEncoder(..., return_hidden_state=True) while TRAINING (reference models/clip/text_encoder.py:125-127: ln_final over every token)
    returns the hidden states attached to the graph.  Fixture: tests/golden/make_golden_text_hidden_grad.py -> text_hidden_grad.npz."""
    from multimodal_amd.models.clip import CLIPTextEncoder

    z = golden("text_hidden_grad.npz")
    enc = CLIPTextEncoder(embedding_dim=64, context_length=77, vocab_size=1000, width=128, dim_feedforward=256, heads=2, layers=2)
    enc.load_state_dict({k[len("sd."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}, strict=True)
    enc = enc.cuda().train()
    hidden = enc(torch.from_numpy(z["ids"]).cuda(), return_hidden_state=True)
    assert hidden.shape == (4, 77, 128) and hidden.grad_fn is not None
    assert np.abs(host(hidden) - z["hidden"]).max() <= 2e-2 * max(1.0, np.abs(z["hidden"]).max())  # (the bound _check puts on a stack's output)
    (hidden * torch.from_numpy(z["w"]).cuda()).sum().backward()
    gscale = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("g."))
    for k, p in enc.named_parameters():
        if "g." + k not in z.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k  # the projection is not on this path
            continue
        ref = z["g." + k].astype(np.float64)
        got = host(p.grad)
        if np.abs(ref).max() < 1e-6 * gscale:  # (in_proj key-bias third: mathematically zero)
            assert np.abs(got).max() <= 1e-3 * gscale, k
            continue
        if k == "token_embedding.weight" or k.endswith("in_proj_bias"):
            # rows of unused tokens / the key third of the packed bias are zero on both sides: compare on the tensor's own scale
            assert np.abs(got - ref).max() <= 6e-2 * np.abs(ref).max(), k
            continue
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        rms = np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-12)
        assert rel <= 6e-2 and rms <= 3e-2, (k, rel, rms)


def test_post_norm_decoder_trains_like_the_reference(golden):
    """r05: POST-NORM TransformerDecoder layers in training -- the reference's default, norm_first=False (modules/layers/transformer.py:289,435-470):
    LayerNorms at the end of the self-attention / cross-attention / feed-forward blocks (DecoderStackFn, layer spec "post").  Fixture:
    tests/golden/make_golden_decoder_post_grad.py -> decoder_post_grad.npz: two layers with cross-attention to 64-wide encoder states (gradient of the
    input, of the encoder states and of every parameter), and one self-attention-only layer behind a final LayerNorm; also with every hidden state
    returned (one autograd node per layer)."""
    from multimodal_amd.modules.layers.transformer import TransformerDecoder

    z = golden("decoder_post_grad.npz")
    causal = torch.ones(9, 9, dtype=torch.bool).tril().cuda()
    for tag, kw in (("dec", dict(n_layer=2, use_cross_attention=True, dim_kv=64)), ("self", dict(n_layer=1, use_cross_attention=False, final_layer_norm_eps=1e-5))):
        for per_layer in (False, True):
            dec = _load(TransformerDecoder(d_model=128, n_head=2, dim_feedforward=256, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=False,
                                           **kw), z, tag)
            x = torch.from_numpy(z[f"{tag}.x"]).cuda().requires_grad_(True)
            enc = torch.from_numpy(z[f"{tag}.enc"]).cuda().requires_grad_(True) if tag == "dec" else None
            out = dec(x, enc, attention_mask=causal, return_hidden_states=per_layer)
            y = out.last_hidden_state
            (y * torch.from_numpy(z[f"{tag}.w"]).cuda()).sum().backward()
            if per_layer:
                assert len(out.hidden_states) == kw["n_layer"] + 1 and all(h.grad_fn is not None for h in out.hidden_states[1:])
            _check(z, tag, dec, y, x)
            if enc is not None:
                ref = z["dec.denc"].astype(np.float64)
                assert np.abs(host(enc.grad) - ref).max() <= 6e-2 * np.abs(ref).max()


def test_stand_alone_decoder_layer_with_cross_attention_mask_trains_like_the_reference(golden):
    """r05: a stand-alone TransformerDecoderLayer hands `cross_attention_mask` to its cross-attention block (reference modules/layers/transformer.py:366-376),
    in training as well (DecoderStackConfig.cross_mask: the forward and backward cross-attention kernels take the same [S, Sk] mask).  Fixture:
    tests/golden/make_golden_decoder_xmask_grad.py -> decoder_xmask_grad.npz, pre- and post-norm, a per-sample boolean [2, 1, 9, 5] mask."""
    from multimodal_amd.modules.layers.transformer import TransformerDecoderLayer

    z = golden("decoder_xmask_grad.npz")
    causal = torch.ones(9, 9, dtype=torch.bool).tril().cuda()
    xmask = torch.from_numpy(z["xmask"]).cuda()
    for tag, nf in (("pre", True), ("post", False)):
        layer = _load(TransformerDecoderLayer(d_model=128, n_head=2, dim_feedforward=256, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=nf,
                                              use_cross_attention=True, dim_kv=64), z, tag)
        x = torch.from_numpy(z[f"{tag}.x"]).cuda().requires_grad_(True)
        enc = torch.from_numpy(z[f"{tag}.enc"]).cuda().requires_grad_(True)
        y, _ = layer(x, enc, attention_mask=causal, cross_attention_mask=xmask)
        (y * torch.from_numpy(z[f"{tag}.w"]).cuda()).sum().backward()
        _check(z, tag, layer, y, x)
        ref = z[f"{tag}.denc"].astype(np.float64)
        assert np.abs(host(enc.grad) - ref).max() <= 6e-2 * np.abs(ref).max()
        # the differentiable forward equals the inference forward with the same mask
        layer.eval()
        with torch.no_grad():
            y_inf, _ = layer(x.detach(), enc.detach(), attention_mask=causal, cross_attention_mask=xmask)
        assert (y_inf - y.detach()).abs().max().item() <= 2e-2 * max(1.0, float(y.detach().abs().max()))


def test_bias_free_attention_projections_train_like_the_reference(golden):
    """`add_bias=False` attention projections in training (raised until r06; reference modules/layers/multi_head_attention.py:107-113 and
    modules/layers/attention.py:100-113): a TransformerDecoderLayer whose self- and cross-attention are MultiHeadAttentionWithCache(add_bias=False)
    (post-norm, causal mask, 64-wide memory) and a FLAVA TransformerEncoderLayer with a bias-free MultiHeadAttention -- output, input gradients and every
    parameter gradient against the reference's autograd (tests/golden/make_golden_nobias_grad.py -> nobias_grad.npz)."""
    from multimodal_amd.models.flava.transformer import TransformerEncoderLayer as FlavaLayer
    from multimodal_amd.modules.layers.attention import MultiHeadAttention, SelfAttention
    from multimodal_amd.modules.layers.multi_head_attention import MultiHeadAttentionWithCache
    from multimodal_amd.modules.layers.transformer import TransformerDecoderLayer

    z = golden("nobias_grad.npz")
    dec = TransformerDecoderLayer(d_model=128, n_head=2, dim_feedforward=256, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=False,
                                  use_cross_attention=True, dim_kv=64)
    dec.attention = MultiHeadAttentionWithCache(dim_q=128, dim_kv=128, num_heads=2, add_bias=False)
    dec.cross_attention = MultiHeadAttentionWithCache(dim_q=128, dim_kv=64, num_heads=2, add_bias=False)
    dec = _load(dec, z, "dec")
    assert dec.attention.q_proj.bias is None and "attention.q_proj.bias" not in dec.state_dict()
    x = torch.from_numpy(z["dec.x"]).cuda().requires_grad_(True)
    enc = torch.from_numpy(z["dec.enc"]).cuda().requires_grad_(True)
    y, _ = dec(x, enc, attention_mask=torch.ones(9, 9, dtype=torch.bool).tril().cuda())
    (y * torch.from_numpy(z["dec.w"]).cuda()).sum().backward()
    _check(z, "dec", dec, y, x)
    ref = z["dec.denc"].astype(np.float64)
    assert np.abs(host(enc.grad) - ref).max() <= 6e-2 * np.abs(ref).max()

    fl = FlavaLayer(d_model=128, n_head=2, dim_feedforward=256, activation=torch.nn.GELU, layer_norm_eps=1e-5, norm_first=True)
    fl.attention = MultiHeadAttention(dim_q=128, dim_kv=128, n_head=2, attn_module=SelfAttention(0.0), add_bias=False)
    fl = _load(fl, z, "flava")
    assert fl.attention.query.bias is None
    y, x = _step(fl, z, "flava", lambda m, t: m(t))
    _check(z, "flava", fl, y, x)
