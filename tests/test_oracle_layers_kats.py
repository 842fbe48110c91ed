"""Pin the generic-layer part of the numpy oracle to the hard-coded expectations of the reference's own layer tests
(tests/modules/layers/test_transformer.py: constant-1 parameters, nn.ReLU feed-forward, eps 1e-12).  CPU-only."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as oc


def _const_sd(module):
    with torch.no_grad():
        for p in module.parameters():
            p.fill_(1.0)
    return {k: v.numpy() for k, v in module.state_dict().items()}


@pytest.mark.parametrize("norm_first,expected", [(True, [[[15.0, 16.0], [18.0, 16.0]]]), (False, [[[0.0, 2.0], [2.0, 0.0]]])])
def test_encoder_layer_kat(norm_first, expected):
    """reference test_transformer.py:22-52."""
    from multimodal_amd.modules.layers.transformer import TransformerEncoderLayer

    sd = _const_sd(TransformerEncoderLayer(d_model=2, n_head=1, dim_feedforward=2, norm_first=norm_first))
    y = oc.layers_encoder_layer(np.array([[[1.0, 2.0], [4.0, 2.0]]], dtype=np.float32), sd, "", 1, 1e-12, norm_first, activation=oc.relu)
    np.testing.assert_allclose(y, expected, atol=1e-4)


@pytest.mark.parametrize("norm_first,last,hidden", [
    (True, [[[30.0, 31.0], [29.0, 30.0]]], [[[[16.0, 17.0], [15.0, 16.0]]], [[[30.0, 31.0], [29.0, 30.0]]]]),
    (False, [[[0.0, 2.0], [0.0, 2.0]]], [[[[0.0, 2.0], [0.0, 2.0]]], [[[0.0, 2.0], [0.0, 2.0]]]]),
])
def test_encoder_kats(norm_first, last, hidden):
    """reference test_transformer.py:60-160 (two layers, hidden states, and the final-LayerNorm variant)."""
    from multimodal_amd.modules.layers.transformer import TransformerEncoder

    x = np.array([[[2.0, 3.0], [1.0, 2.0]]], dtype=np.float32)
    sd = _const_sd(TransformerEncoder(n_layer=2, d_model=2, n_head=1, dim_feedforward=2, norm_first=norm_first))
    y, hs = oc.layers_encoder(x, sd, "", 1, 1e-12, norm_first, activation=oc.relu)
    np.testing.assert_allclose(y, last, atol=1e-4)
    assert np.array_equal(hs[0], x)
    for got, want in zip(hs[1:], hidden):
        np.testing.assert_allclose(got, want, atol=1e-4)
    sd = _const_sd(TransformerEncoder(n_layer=2, d_model=2, n_head=1, dim_feedforward=2, norm_first=norm_first, final_layer_norm_eps=1e-5))
    y, _ = oc.layers_encoder(x, sd, "", 1, 1e-12, norm_first, final_eps=1e-5, activation=oc.relu)
    want = [[[1.9073e-05, 2.0], [2.2888e-05, 2.0]]] if norm_first else [[[5.0068e-06, 2.0], [5.0068e-06, 2.0]]]
    np.testing.assert_allclose(y, want, atol=1e-4)


@pytest.mark.parametrize("norm_first,expected", [(True, [[[15.0, 16.0], [18.0, 16.0], [15.0, 15.0]]]), (False, [[[0.0, 2.0], [2.0, 0.0], [1.0, 1.0]]])])
def test_decoder_layer_without_cross_attention_kat(norm_first, expected):
    """reference test_transformer.py:244-262."""
    from multimodal_amd.modules.layers.transformer import TransformerDecoderLayer

    sd = _const_sd(TransformerDecoderLayer(d_model=2, n_head=1, dim_feedforward=2, norm_first=norm_first, use_cross_attention=False))
    y = oc.layers_decoder_layer(np.array([[[1.0, 2.0], [4.0, 2.0], [1.0, 1.0]]], dtype=np.float32), None, sd, "", 1, 1e-12, norm_first=norm_first,
                                activation=oc.relu)
    np.testing.assert_allclose(y, expected, atol=1e-3)
