"""CoCa on an MI355X: the general attention kernel and the drop-in modules against outputs of the
reference itself (tests/golden/make_golden_coca.py) and the numpy oracle.  Tolerances as for CLIP / FLAVA: bf16 MFMA operands,
fp32 accumulation and residual stream — L2-normalised pooled outputs |d| <= 4e-3, vocabulary logits |d| <= 3e-2, losses <= 1e-2.
"""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as oc
from tests._util import assert_checksums, sd_to_numpy
from tests.conftest import set_rng_seed
from tests.golden.make_golden import seed
from tests.golden.make_golden_coca import POOL96, randomize, SMALL

pytestmark = pytest.mark.gpu

EMB_TOL, LOGIT_TOL, LOSS_TOL, HID_TOL = 4e-3, 3e-2, 1e-2, 3e-2


@pytest.fixture(scope="module", autouse=True)
def _built():
    from multimodal_amd import build

    build.build()


def host(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("B,Sq,Sk,H,hd,causal,kmask,fmask,shared", [
    (2, 12, 12, 2, 64, True, False, False, False), (3, 76, 256, 12, 64, False, False, False, False),
    (2, 257, 256, 8, 96, False, False, False, True), (2, 1, 256, 8, 96, False, False, False, True),
    (2, 77, 77, 3, 64, False, False, True, False), (2, 77, 77, 2, 96, True, True, False, False),
    (1, 40, 288, 1, 96, False, True, True, False), (2, 50, 49, 8, 64, False, False, False, True)])
def test_attention_x_kernel(B, Sq, Sk, H, hd, causal, kmask, fmask, shared):
    from multimodal_amd import ops

    set_rng_seed(Sq * 7 + Sk)
    D = H * hd
    qrows = Sq if shared else B * Sq
    wide = torch.randn(qrows, D + 64).to(torch.bfloat16)  # q is a column-slice view of a wider matrix
    kv = torch.randn(B * Sk, 2 * D).to(torch.bfloat16)
    km = fm = None
    if kmask:
        km = (torch.rand(B, Sk) > 0.3).to(torch.uint8)
        km[:, 0] = 1
    if fmask:
        fm = (torch.rand(B, Sq, Sk) > 0.4).to(torch.uint8)
        fm[:, :, 0] = 1
    wg, kvg = wide.cuda(), kv.cuda()
    mask = ops.AttnMask(causal=causal, key_mask=km.cuda() if km is not None else None, full=fm.cuda() if fm is not None else None)
    out, probs = ops.attention_x_fwd(wg[:, 64:], kvg[:, :D], kvg[:, D:], B, Sq, Sk, H, hd, mask, shared_q=shared, want_probs=True)
    q = wide[:, 64:].float().numpy().astype(np.float64)
    q = np.broadcast_to(q.reshape(1, Sq, H, hd), (B, Sq, H, hd)) if shared else q.reshape(B, Sq, H, hd)
    k = kv[:, :D].float().numpy().astype(np.float64).reshape(B, Sk, H, hd)
    v = kv[:, D:].float().numpy().astype(np.float64).reshape(B, Sk, H, hd)
    s = np.einsum("bqhd,bkhd->bhqk", q, k) / np.sqrt(hd)
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
    o = np.einsum("bhqk,bkhd->bqhd", p, v).reshape(B * Sq, D)
    assert np.abs(host(probs) - p).max() <= 2e-6
    assert np.abs(host(out) - o).max() <= 2e-2 * max(1.0, np.abs(o).max())
    out2, none = ops.attention_x_fwd(wg[:, 64:], kvg[:, :D], kvg[:, D:], B, Sq, Sk, H, hd, mask, shared_q=shared)
    assert none is None and torch.equal(out2, out)


def test_coca_text_embed_and_mask_kernels():
    from multimodal_amd import ops

    set_rng_seed(8)
    B, S, d, vocab = 5, 76, 768, 300
    ids = torch.randint(0, vocab, (B, S))
    ids[1, 40:] = 0
    ids[3, 1:] = 0
    table, pos, cls = torch.randn(vocab, d), torch.randn(S + 1, d), torch.randn(d)
    x = ops.coca_text_embed(ids.cuda(), table.cuda(), pos.cuda(), cls.cuda()).cpu().view(B, S + 1, d)
    ref = torch.cat([table[ids], cls.reshape(1, 1, -1).repeat(B, 1, 1)], 1) + pos
    assert torch.equal(x, ref)
    x2 = ops.coca_text_embed(ids.cuda(), table.cuda(), pos.cuda(), None).cpu().view(B, S, d)
    assert torch.equal(x2, table[ids] + pos[:S])
    ref_mask = oc.coca_text_mask(ids.numpy(), 0)[:, 0]
    assert np.array_equal(ops.coca_text_mask(ids.cuda(), pad_id=0).cpu().numpy().astype(bool), ref_mask)
    for pm in ((ids != 0), (ids != 0).float(), (ids != 0).long()):
        assert np.array_equal(ops.coca_text_mask(pm.cuda()).cpu().numpy().astype(bool), ref_mask)


def rebuild(kw, cascaded, seed_v, z, prefix):
    from multimodal_amd.models.coca.coca_model import coca_vit

    seed(seed_v)
    model = coca_vit(**kw, cascaded_pooler=cascaded).eval()
    randomize(model, torch.Generator().manual_seed(seed_v + 1))
    assert_checksums(model, {"keys": z[prefix + "keys"], "sums": z[prefix + "sums"], "asums": z[prefix + "asums"]})
    return model


@pytest.mark.parametrize("fixture,kw,cascaded,seed_v,prefix", [
    ("coca_small.npz", SMALL, False, 51, "par."), ("coca_small.npz", SMALL, True, 52, "cas."), ("coca_pool96.npz", POOL96, False, 53, "par.")])
def test_small_coca_vs_reference_fixture(golden, fixture, kw, cascaded, seed_v, prefix):
    from multimodal_amd import ops
    from multimodal_amd.models.coca.coca_model import CoCaForPretraining, MultimodalOutput

    z = golden(fixture)
    model = rebuild(kw, cascaded, seed_v, z, prefix).cuda()
    images, texts = torch.from_numpy(z[prefix + "images"]).cuda(), torch.from_numpy(z[prefix + "texts"]).cuda()
    with torch.no_grad():
        out = model(images, texts)
        out_pm = model(images, texts, texts != 0)  # explicit padding mask == the default (ids != pad)
    assert isinstance(out, MultimodalOutput) and out.multimodal_pooled_embeddings is None
    rep = {}
    for k, tol in (("image_pooled_output", EMB_TOL), ("text_pooled_output", EMB_TOL), ("multimodal_embeddings", LOGIT_TOL)):
        got = host(getattr(out, k))
        assert got.shape == z[prefix + k].shape, (k, got.shape)  # cascaded: image_pooled_output stays [B, 1, D]
        rep[k] = np.abs(got - z[prefix + k]).max()
        assert rep[k] <= tol, (k, rep[k])
    assert torch.equal(out_pm.text_pooled_output, out.text_pooled_output)
    assert torch.equal(out_pm.multimodal_embeddings, out.multimodal_embeddings)
    pre = CoCaForPretraining(model).cuda().eval()
    if cascaded:
        with pytest.raises(ops.MmamdError, match="cascaded"), torch.no_grad():
            pre(images, texts)
    else:
        with torch.no_grad():
            losses = pre(images, texts)
        rep["contrastive"] = abs(float(losses["contrastive"]) - float(z[prefix + "loss_contrastive"]))
        rep["captioning"] = abs(float(losses["captioning"]) - float(z[prefix + "loss_captioning"]))
        assert rep["contrastive"] <= LOSS_TOL and rep["captioning"] <= LOSS_TOL, rep
    print(f"coca {fixture}:{prefix} parity |d|:", {k: float(f"{v:.2e}") for k, v in rep.items()})


def test_coca_layers_module_api_vs_oracle():
    """The generic layers CoCa is built from, called directly like the reference's unit tests."""
    from multimodal_amd import ops
    from multimodal_amd.modules.layers.attention_pooler import AttentionPooler, CascadedAttentionPooler
    from multimodal_amd.modules.layers.multi_head_attention import MultiHeadAttentionWithCache, MultiHeadSelfAttention
    from multimodal_amd.modules.layers.transformer import TransformerDecoder, TransformerDecoderLayer, TransformerEncoder, TransformerEncoderLayer

    set_rng_seed(31)
    B, S, Sk, d = 3, 10, 7, 128
    x, enc = torch.randn(B, S, d), torch.randn(B, Sk, 192)
    G = torch.nn.GELU
    with torch.no_grad():
        # self-attention with a boolean mask, and causal
        sa = MultiHeadSelfAttention(d, 2).cuda().eval()
        sd = sd_to_numpy(sa)
        m = torch.rand(B, 1, S, S) > 0.3
        m[..., 0] = True
        assert np.abs(host(sa(x.cuda(), attn_mask=m.cuda())) - oc.mh_self_attention(x.numpy(), sd, "", 2, m.numpy())).max() <= 2e-2
        assert np.abs(host(sa(x.cuda(), is_causal=True)) - oc.mh_self_attention(x.numpy(), sd, "", 2, None, True)).max() <= 2e-2
        # cross-attention, different kv width
        ca = MultiHeadAttentionWithCache(d, 192, 2).cuda().eval()
        kv = enc.cuda()
        assert np.abs(host(ca(x.cuda(), kv, kv)) - oc.mha_with_cache(x.numpy(), enc.numpy(), sd_to_numpy(ca), "", 2)).max() <= 2e-2
        cached = ca(x.cuda(), kv, kv, use_cache=True)  # cross-attention with use_cache: output + this call's keys / values
        assert np.abs(host(cached.attn_output) - oc.mha_with_cache(x.numpy(), enc.numpy(), sd_to_numpy(ca), "", 2)).max() <= 2e-2
        assert cached.past_key_value[0].shape == (x.shape[0], 2, enc.shape[1], 64)
        # encoder layers, pre- and post-norm; encoder with hidden states + final LN
        for nf in (True, False):
            el = TransformerEncoderLayer(d, 2, 256, activation=G, layer_norm_eps=1e-5, norm_first=nf).cuda().eval()
            ref = oc.layers_encoder_layer(x.numpy(), sd_to_numpy(el), "", 2, 1e-5, nf)
            assert np.abs(host(el(x.cuda())) - ref).max() <= HID_TOL, nf
        te = TransformerEncoder(2, d, 2, 256, activation=G, layer_norm_eps=1e-5, norm_first=True, final_layer_norm_eps=1e-6).cuda().eval()
        o = te(x.cuda(), return_hidden_states=True)
        last, hidden = oc.layers_encoder(x.numpy(), sd_to_numpy(te), "", 2, 1e-5, True, 1e-6)
        assert len(o.hidden_states) == 3 and np.abs(host(o.last_hidden_state) - last).max() <= HID_TOL
        assert np.abs(host(o.hidden_states[-1]) - hidden[-1]).max() <= HID_TOL and te(x.cuda()).hidden_states is None
        # decoder layer with cross-attention (dim_kv 192), causal boolean mask; decoder stack
        dl = TransformerDecoderLayer(d, 2, 256, activation=G, layer_norm_eps=1e-5, norm_first=True, dim_kv=192).cuda().eval()
        cm = torch.tril(torch.ones(S, S)).bool()
        y, kvc = dl(x.cuda(), kv, attention_mask=cm.cuda())
        ref = oc.layers_decoder_layer(x.numpy(), enc.numpy(), sd_to_numpy(dl), "", 2, 1e-5, attend=cm.numpy())
        assert kvc is None and np.abs(host(y) - ref).max() <= HID_TOL
        td = TransformerDecoder(2, d, 2, 256, activation=G, layer_norm_eps=1e-5, norm_first=True, dim_kv=192, final_layer_norm_eps=1e-5).cuda().eval()
        o = td(x.cuda(), kv, attention_mask=cm.cuda())
        ref = oc.layers_decoder(x.numpy(), enc.numpy(), sd_to_numpy(td), "", 2, 1e-5, attend=cm.numpy(), final_eps=1e-5)
        assert np.abs(host(o.last_hidden_state) - ref).max() <= HID_TOL and o.current_key_values == []
        # attention poolers: 64- and 96-wide heads, cascaded
        for (din, dout, h, nq) in ((192, 128, 2, 5), (128, 192, 2, 257)):
            ap = AttentionPooler(din, dout, h, n_queries=nq).cuda().eval()
            xin = torch.randn(B, 33, din)
            ref = oc.attention_pooler(xin.numpy(), sd_to_numpy(ap), "", h)
            got = ap(xin.cuda())
            assert got.shape == (B, nq, dout) and np.abs(host(got) - ref).max() <= 2e-2
        cp = CascadedAttentionPooler([AttentionPooler(128, 128, 2, n_queries=4), AttentionPooler(128, 128, 2, n_queries=1)]).cuda().eval()
        outs = cp(torch.randn(B, 9, 128).cuda())
        assert [tuple(t.shape) for t in outs] == [(B, 4, 128), (B, 1, 128)]
        with pytest.raises(ops.MmamdError, match="64- and 96-wide"):
            MultiHeadSelfAttention(8, 2).cuda().eval()(torch.randn(1, 4, 8).cuda())  # the reference's KAT sizes are not kernel-legal


def test_vision_transformer_with_cls_and_patch14():
    from multimodal_amd.modules.encoders.vision_transformer import vision_transformer

    set_rng_seed(2)
    for (patch, size, cls) in ((16, 64, True), (14, 56, False)):
        vit = vision_transformer(patch_size=patch, hidden_dim=128, dim_feedforward=256, n_layer=1, n_head=2, image_size=size,
                                 include_cls_embed=cls, layer_norm_eps=1e-5, final_layer_norm_eps=1e-5).eval()
        with torch.no_grad():
            for p in vit.parameters():
                p.add_(torch.randn_like(p) * 0.02)
        sd = sd_to_numpy(vit)
        img = torch.randn(2, 3, size, size)
        with torch.no_grad():
            o = vit.cuda()(img.cuda())
        x = oc.layers_patch_embeddings(img.numpy(), sd, "embeddings.")
        last, hidden = oc.layers_encoder(x, sd, "encoder.", 2, 1e-5, True, 1e-5)
        assert np.abs(host(o.hidden_states[0]) - x).max() <= 2e-2  # patch-embedding GEMM from bf16 operands
        assert np.abs(host(o.last_hidden_state) - last).max() <= HID_TOL and o.pooler_output is None


@torch.no_grad()  # inference contract: eval-mode forwards with autograd recording raise (tests/test_host_api_*.py)
def test_key_value_cache_vs_reference_fixture(golden):
    """MultiHeadAttentionWithCache / TransformerDecoder with past_key_value(s) and use_cache (reference
    modules/layers/multi_head_attention.py:158-179, transformer.py:336-359,586-657): cached + new keys in one attention call, the
    returned caches in the reference's [B, H, S, hd] shape, and incremental decoding == the full causal pass."""
    from multimodal_amd.modules.layers.multi_head_attention import MHAWithCacheOutput, MultiHeadAttentionWithCache
    from multimodal_amd.modules.layers.transformer import TransformerDecoder

    z = golden("kv_cache.npz")
    sub = lambda pre: {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
    mha = MultiHeadAttentionWithCache(dim_q=128, dim_kv=128, num_heads=2)
    mha.load_state_dict(sub("mha.sd."), strict=True)
    mha = mha.cuda().eval()
    x = torch.from_numpy(z["mha.x"]).cuda()
    pk, pv, mask = (torch.from_numpy(z[k]).cuda() for k in ("mha.pk", "mha.pv", "mha.mask"))
    with torch.no_grad():
        o = mha(x, x, x, attn_mask=mask, past_key_value=(pk, pv), use_cache=True)
        assert isinstance(o, MHAWithCacheOutput) and o.past_key_value[0].shape == (3, 2, 8, 64)
        assert np.abs(host(o.attn_output) - z["mha.out"]).max() <= 3e-2
        assert np.abs(host(o.past_key_value[0]) - z["mha.key"]).max() <= 2.0 ** -7 * np.abs(z["mha.key"]).max()   # held in bf16
        assert np.abs(host(o.past_key_value[1]) - z["mha.value"]).max() <= 2.0 ** -7 * np.abs(z["mha.value"]).max()
        o2 = mha(x, x, x, use_cache=True)
        assert np.abs(host(o2.attn_output) - z["mha.out_nopast"]).max() <= 3e-2 and o2.past_key_value[0].shape == (3, 2, 3, 64)
        plain = mha(x, x, x, attn_mask=mask, past_key_value=(pk, pv))  # past without use_cache: just the tensor, like the reference
        assert torch.equal(plain, o.attn_output)

    dec = TransformerDecoder(n_layer=2, d_model=128, n_head=2, dim_feedforward=256, activation=torch.nn.GELU, layer_norm_eps=1e-5,
                             norm_first=True, use_cross_attention=True, dim_kv=128, final_layer_norm_eps=1e-5)
    dec.load_state_dict(sub("dec.sd."), strict=True)
    dec = dec.cuda().eval()
    h, enc = torch.from_numpy(z["dec.h"]).cuda(), torch.from_numpy(z["dec.enc"]).cuda()
    causal = torch.ones(6, 6, dtype=torch.bool).tril().cuda()
    with torch.no_grad():
        full = dec(h, enc, attention_mask=causal)
        assert np.abs(host(full.last_hidden_state) - z["dec.full"]).max() <= 4e-2 and full.current_key_values == []
        o = dec(h[:, :4], enc, attention_mask=causal[:4, :4], use_cache=True)
        steps, cache = [o.last_hidden_state], o.current_key_values
        assert len(cache) == 2 and cache[0][0].shape == (2, 2, 4, 64)
        for t in (4, 5):
            o = dec(h[:, t:t + 1], enc, attention_mask=causal[t:t + 1, :t + 1], past_key_values=cache, use_cache=True)
            steps.append(o.last_hidden_state)
            cache = o.current_key_values
    inc = torch.cat(steps, dim=1)
    assert np.abs(host(inc) - z["dec.full"]).max() <= 4e-2
    assert np.abs(host(inc) - host(full.last_hidden_state)).max() <= 2e-2   # incremental == full on the GPU path
    assert np.abs(host(cache[1][0]) - z["dec.cache_k1"]).max() <= 3e-2 and np.abs(host(cache[0][1]) - z["dec.cache_v0"]).max() <= 3e-2
    with pytest.raises(ValueError):
        dec(h[:, 4:5], enc, past_key_values=cache[:1], use_cache=True)
