"""The numpy oracle's CoCa restatement (oracle/clip_oracle.py: coca_*) pinned to the reference: its constant-weight KAT
(tests/models/coca/test_coca_model.py: 0.3536 / 8.0 / 0.6931 / 3.9120) and outputs of the reference itself on seeded random
small models (tests/golden/make_golden_coca.py).  Weights of the random models are re-created from the seed through the drop-in
modules and verified against the reference's per-tensor checksums."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as oc
from tests._util import assert_checksums, fixture_sd, sd_to_numpy
from tests.golden.make_golden import seed
from tests.golden.make_golden_coca import POOL96, randomize, SMALL


def rebuild(kw, cascaded, seed_v, z, prefix):
    from multimodal_amd.models.coca.coca_model import coca_vit

    seed(seed_v)
    model = coca_vit(**kw, cascaded_pooler=cascaded).eval()
    randomize(model, torch.Generator().manual_seed(seed_v + 1))
    assert_checksums(model, {"keys": z[prefix + "keys"], "sums": z[prefix + "sums"], "asums": z[prefix + "asums"]})
    return model


def test_coca_kat_constant_weights(golden):
    z = golden("coca_kat.npz")
    sd = fixture_sd(z)
    out = oc.coca_model_forward(sd, z["images"], z["texts"], 2, 2, 2, 2, cascaded=False)
    assert np.abs(out["image_pooled_output"] - 0.3536).max() <= 1e-4 and np.abs(out["text_pooled_output"] - 0.3536).max() <= 1e-4
    assert np.abs(out["multimodal_embeddings"] - 8.0).max() <= 1e-4
    for k in ("image_pooled_output", "text_pooled_output", "multimodal_embeddings"):
        assert np.abs(out[k] - z[k]).max() <= 1e-5, k
    losses = oc.coca_pretraining_losses(out, z["texts"], float(z["logit_scale"]))
    assert abs(float(losses["contrastive"]) - 0.6931) <= 1e-4 and abs(float(losses["captioning"]) - 3.9120) <= 1e-4


@pytest.mark.parametrize("fixture,kw,cascaded,seed_v,prefix,heads", [
    ("coca_small.npz", SMALL, False, 51, "par.", (2, 2, 2, 2)), ("coca_small.npz", SMALL, True, 52, "cas.", (2, 2, 2, 2)),
    ("coca_pool96.npz", POOL96, False, 53, "par.", (2, 3, 3, 2))])
def test_coca_oracle_vs_reference_outputs(golden, fixture, kw, cascaded, seed_v, prefix, heads):
    z = golden(fixture)
    model = rebuild(kw, cascaded, seed_v, z, prefix)
    out = oc.coca_model_forward(sd_to_numpy(model), z[prefix + "images"], z[prefix + "texts"], *heads, cascaded=cascaded)
    for k, tol in (("image_pooled_output", 2e-6), ("text_pooled_output", 2e-6), ("multimodal_embeddings", 5e-5)):
        assert out[k].shape == z[prefix + k].shape, k
        assert np.abs(out[k] - z[prefix + k]).max() <= tol, (k, np.abs(out[k] - z[prefix + k]).max())
    if not cascaded:
        losses = oc.coca_pretraining_losses(out, z[prefix + "texts"], np.log(1 / 0.07))
        assert abs(float(losses["contrastive"]) - float(z[prefix + "loss_contrastive"])) <= 2e-5
        assert abs(float(losses["captioning"]) - float(z[prefix + "loss_captioning"])) <= 2e-5


def test_coca_text_mask_matches_reference_build_mask():
    """build_mask semantics restated independently with torch ops (F.pad shifts the padding mask by one column)."""
    import torch.nn.functional as F

    ids = torch.tensor([[5, 3, 0, 0], [1, 2, 3, 4], [0, 7, 0, 9]])
    pm = (ids != 0).unsqueeze(1)
    ref = (F.pad(pm, (1, 0, pm.shape[2], 0), value=1.0) * torch.tril(torch.ones(5, 5)).bool()).unsqueeze(1)
    assert np.array_equal(oc.coca_text_mask(ids.numpy(), 0), ref.bool().numpy())


@pytest.mark.parametrize("pad_idx,expected_pooled", [
    (0, [[5.5019, -4.5114, 3.0416], [3.4487, -6.2877, 3.1439]]),
    (None, [[5.5019, -4.5114, 3.0416], [3.4142, -6.3097, 3.1282]]),
])
def test_reference_text_decoder_kats(pad_idx, expected_pooled):
    """tests/models/coca/test_text_decoder.py:186-252 of the reference (embed_cls=True rows): all parameters 1.0
    (init_weights_with_constant), then seed 0 and nn.init.normal_ on the token embeddings, text projection, the four attention
    projections of both layers and the CLS embedding, in that order; pooled output and per-token means are hard-coded there."""
    import torch
    from torch import nn

    from multimodal_amd.models.coca.text_decoder import CoCaTextDecoder
    from tests.conftest import set_rng_seed

    dec = CoCaTextDecoder(vocab_size=12, num_positions=6, embedding_dim=8, n_layer=2, n_head=2, dim_feedforward=32, output_dim=3,
                          pad_idx=pad_idx, embed_cls=True)
    with torch.no_grad():
        for p_ in dec.parameters():
            p_.fill_(1.0)
    set_rng_seed(0)
    nn.init.normal_(dec.embeddings.token_embeddings.weight)
    nn.init.normal_(dec.text_projection.weight)
    for block in dec.transformer_decoder.layer:
        for name in ("q_proj", "k_proj", "v_proj", "output_proj"):
            nn.init.normal_(getattr(block.attention, name).weight)
    nn.init.normal_(dec.embeddings.cls_embedding)
    sd = {k: v.detach().numpy() for k, v in dec.state_dict().items()}
    ids = np.array([[2, 4, 5, 7, 9, 1], [6, 8, 1, 0, 0, 0]])
    pooled, tokens = oc.coca_text_decoder(sd, "", ids, heads=2, pad_idx=pad_idx)
    np.testing.assert_allclose(pooled, expected_pooled, atol=1e-3)
    assert tokens.shape == (2, 5, 8)
    np.testing.assert_allclose(tokens.mean(-1), [[585.0038, 587.7021, 588.5288, 585.5997, 588.6697], [586.2949, 585.1484, 588.0995, 590.9081, 591.0029]],
                               atol=2e-3)


def test_reference_multimodal_decoder_kat():
    """tests/models/coca/test_multimodal_decoder.py:30-112 of the reference: all parameters 1.0, then arange weights in the last MLP
    Linear, the output projection and the final LayerNorm; text = arange(0,1,1/40), image = arange(10,20,1/8); every output row is
    [58.2492, 66.7214, 75.1935]."""
    import torch
    from torch import nn

    from multimodal_amd.models.coca.multimodal_decoder import CoCaMultimodalDecoder

    dec = CoCaMultimodalDecoder(input_seq_len=5, text_embedding_dim=4, n_layer=2, n_head=2, dim_feedforward=16, output_dim=3, final_layer_norm_eps=1e-5)
    with torch.no_grad():
        for p_ in dec.parameters():
            p_.fill_(1.0)
    last = dec.transformer_decoder.layer[1].feedforward.model[2]
    last.weight = nn.Parameter(torch.arange(last.weight.numel(), dtype=torch.float).reshape(last.weight.shape))
    dec.output_projection.weight = nn.Parameter(torch.arange(dec.output_projection.weight.numel(), dtype=torch.float).reshape(dec.output_projection.weight.T.shape).T)
    fln = dec.transformer_decoder.final_layer_norm
    fln.weight = nn.Parameter(torch.arange(fln.weight.numel(), dtype=torch.float))
    sd = {k: v.detach().numpy() for k, v in dec.state_dict().items()}
    text = np.arange(0.0, 1.0, 1.0 / 40, dtype=np.float32).reshape(2, 5, 4)
    image = np.arange(10.0, 20.0, 1.0 / 8, dtype=np.float32).reshape(2, 10, 4)
    causal = np.tril(np.ones((5, 5), dtype=bool))
    h = oc.layers_decoder(text, image, sd, "transformer_decoder.", 2, 1e-5, attend=causal, final_eps=1e-5)
    out = h @ sd["output_projection.weight"].T
    np.testing.assert_allclose(out, np.broadcast_to(np.array([58.2492, 66.7214, 75.1935]), (2, 5, 3)), atol=1e-4)


def test_reference_attention_pooler_kats():
    """tests/modules/layers/test_attention_pooler.py:47-112 of the reference: constant-1 parameters, randn inputs (seed 0): output sums
    144 for AttentionPooler(4 -> 6, 2 heads, 12 queries) and [144, 20] for the cascade with a second 1-query pooler (6 -> 10)."""
    import torch

    from multimodal_amd.modules.layers.attention_pooler import AttentionPooler
    from tests.conftest import set_rng_seed

    set_rng_seed(0)
    x = torch.randn(2, 8, 4).numpy()
    p1 = AttentionPooler(input_embed_dim=4, output_embed_dim=6, n_head=2, n_queries=12)
    p2 = AttentionPooler(input_embed_dim=6, output_embed_dim=10, n_head=2, n_queries=1)
    sds = []
    for p_ in (p1, p2):
        with torch.no_grad():
            for q in p_.parameters():
                q.fill_(1.0)
        sds.append({k: v.numpy() for k, v in p_.state_dict().items()})
    y1 = oc.attention_pooler(x, sds[0], "", 2)
    assert y1.shape == (2, 12, 6) and abs(float(y1.sum()) - 144.0) <= 1e-3
    y2 = oc.attention_pooler(y1, sds[1], "", 2)
    assert y2.shape == (2, 1, 10) and abs(float(y2.sum()) - 20.0) <= 1e-3
